// Context, device memory, events, error strings and the pinned rotation data (host shim H1).
#include <cmath>
#include <cstdlib>

#include <string>

#include "pb3d_internal.h"

static int pool_flush(pb3d_ctx* ctx);

static thread_local char g_err[512] = "";

void pb3d_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

int pb3d_version(void) { return 101; }

// number of host waits on the context's stream so far (development / tests: a resident pipeline is meant to wait a bounded number of times)
int64_t pb3d_sync_count(pb3d_ctx* ctx) { return ctx ? (int64_t)ctx->sync_count : -1; }

const char* pb3d_last_error(void) { return g_err; }

int pb3d_device_count(int* n) {
    PB3D_REQUIRE(n != nullptr, "pb3d_device_count: null output");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c = 0;
    }
    *n = c;
    return PB3D_OK;
}

int pb3d_create(int device, pb3d_ctx** out) {
    PB3D_REQUIRE(out != nullptr, "pb3d_create: null output");
    *out = nullptr;
    int n = 0;
    pb3d_device_count(&n);
    if (n <= 0) {
        pb3d_set_error("pb3d_create: no HIP device visible (libpb3d has no CPU fallback)");
        return PB3D_ENODEVICE;
    }
    PB3D_REQUIRE(device >= 0 && device < n, "pb3d_create: device %d out of range (have %d)", device, n);
    PB3D_HIP(hipSetDevice(device));
    pb3d_ctx* ctx = (pb3d_ctx*)calloc(1, sizeof(pb3d_ctx));
    if (!ctx) {
        pb3d_set_error("pb3d_create: out of host memory");
        return PB3D_ENOMEM;
    }
    ctx->device = device;
    {
        auto env_int = [](const char* name) { const char* v = getenv(name); return v ? atoi(v) : 0; };
        ctx->tune_uncap = env_int("PB3D_UNCAP");
        ctx->tune_sliced = env_int("PB3D_SLICED");
        ctx->tune_s32_gpw = env_int("PB3D_S32_GPW");
        ctx->tune_rot90_wide = env_int("PB3D_ROT90_WIDE");
    }
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        free(ctx);
        pb3d_set_error("hipGetDeviceProperties failed: %s", hipGetErrorString(e));
        return PB3D_ENODEVICE;
    }
    ctx->cus = prop.multiProcessorCount;
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        free(ctx);
        pb3d_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        return PB3D_ENODEVICE;
    }
    e = hipEventCreateWithFlags(&ctx->s32_ev, hipEventDisableTiming);
    if (e != hipSuccess) {
        (void)hipStreamDestroy(ctx->stream);
        free(ctx);
        pb3d_set_error("hipEventCreate failed: %s", hipGetErrorString(e));
        return PB3D_ENODEVICE;
    }
    {
        const char* v = getenv("PB3D_DEVICE_POOL_MB");
        ctx->pool_cap = v ? (size_t)(atoll(v) < 0 ? 0 : atoll(v)) << 20 : (size_t)prop.totalGlobalMem / 4;
    }
    ctx->pinned_bytes = 1 << 16;
    e = hipHostMalloc(&ctx->pinned, ctx->pinned_bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipStreamDestroy(ctx->stream);
        free(ctx);
        pb3d_set_error("hipHostMalloc failed: %s", hipGetErrorString(e));
        return PB3D_ENOMEM;
    }
    // PB3D_KNOBS="name=value,name=value": development knobs by name (pb3d_set_tuning), read once here; an unknown name or a value out of
    // range fails the creation loudly rather than silently measuring the wrong kernel
    if (const char* kn = getenv("PB3D_KNOBS")) {
        std::string all(kn);
        size_t pos = 0;
        while (pos < all.size()) {
            const size_t end = all.find(',', pos);
            const std::string item = all.substr(pos, end == std::string::npos ? std::string::npos : end - pos);
            pos = end == std::string::npos ? all.size() : end + 1;
            if (item.empty()) continue;
            const size_t eq = item.find('=');
            const int rc = eq == std::string::npos ? PB3D_EINVAL : pb3d_set_tuning(ctx, item.substr(0, eq).c_str(), atoi(item.c_str() + eq + 1));
            if (rc != PB3D_OK) {
                if (eq == std::string::npos) pb3d_set_error("PB3D_KNOBS: '%s' is not name=value", item.c_str());
                (void)hipHostFree(ctx->pinned);
                (void)hipStreamDestroy(ctx->stream);
                free(ctx);
                return PB3D_EINVAL;
            }
        }
    }
    *out = ctx;
    return PB3D_OK;
}

namespace {
struct Knob { const char* name; int pb3d_ctx::*field; int lo, hi; const char* what; };
const Knob kKnobs[] = {
    {"sliced", &pb3d_ctx::tune_sliced, 0, 1, "0 (rotation steps on 0/1 data run bit-sliced) or 1 (never: byte chain)"},
    {"rot90_wide", &pb3d_ctx::tune_rot90_wide, 0, 2, "0 (choose), 1 or 2 (the 128-tile / 128-byte-segment kernels)"},
    {"rot90_fill", &pb3d_ctx::tune_rot90_fill, 0, 64, "a count of workgroups per CU (0 = the built-in rule)"},
    {"rot90_order", &pb3d_ctx::tune_rot90_order, 0, 1, "0 (a plane chunk per XCD) or 1 (plain x-fastest order)"},
    {"rot90_flat", &pb3d_ctx::tune_rot90_flat, 0, 2, "0 (choose), 1 (row-wise tile kernel) or 2 (flat on whole-line streams only)"},
    {"rot90_mask_block", &pb3d_ctx::tune_rot90_mask_block, 0, 1, "0 (mask bytes once into LDS) or 1 (per plane / segment)"},
    {"s32_gpw", &pb3d_ctx::tune_s32_gpw, 0, 1 << 20, "a count of plane groups"},
    {"s32_order", &pb3d_ctx::tune_s32_order, 0, 1, "0 (plane groups fastest) or 1 (x fastest)"},
    {"s32_fuse_last", &pb3d_ctx::tune_s32_fuse_last, 0, 1, "0 (fused) or 1 (table step)"},
    {"ccl_blocks", &pb3d_ctx::tune_ccl_blocks, 0, 1 << 20, "a count of workgroups per CU"},
    {"ccl_init_blocks", &pb3d_ctx::tune_ccl_init_blocks, 0, 1 << 20, "a count of workgroups per CU"},
    {"ccl_tilecols", &pb3d_ctx::tune_ccl_tilecols, 0, 64, "at most 64"},
    {"ccl_merge", &pb3d_ctx::tune_ccl_merge, 0, 1, "0 (tile kernels) or 1 (pairwise kernel)"},
    {"points_fill", &pb3d_ctx::tune_points_fill, 0, 1, "0 (wave-private fill) or 1 (block form)"},
    {"points_onepass", &pb3d_ctx::tune_points_onepass, 0, 1, "0 (count + fill) or 1 (one-pass look-back form in the host entry)"},
    {"orient_tile", &pb3d_ctx::tune_orient_tile, 0, 1, "0 (128-pixel tiles where they apply) or 1 (never)"},
    {"global_composed", &pb3d_ctx::tune_global_composed, 0, 1, "0 (mask bits -> sliced chain -> colours) or 1 (ones -> process -> colour)"},
    {"per_job", &pb3d_ctx::tune_per_job, 0, 1, "0 (merged / fused forms) or 1 (job by job)"},
    {"no_table_cache", &pb3d_ctx::tune_no_table_cache, 0, 1, "0 (tables cached across calls) or 1 (rebuilt every call)"},
    {"part90_inflight", &pb3d_ctx::tune_part90_inflight, 0, 44, "10 UA + UE with UA, UE in 1, 2, 4 (0 = the default 2 / 4)"},
    {"crop_ablate", &pb3d_ctx::tune_crop_ablate, 0, 1 << 20, "ablation switches of k_crop_chain"},
    {"uncap", &pb3d_ctx::tune_uncap, 0, 1, "0 or 1 (every grid-stride kernel as one workgroup per tile)"},
};
}  // namespace

int pb3d_set_tuning(pb3d_ctx* ctx, const char* name, int value) {
    PB3D_REQUIRE(ctx != nullptr && name != nullptr, "pb3d_set_tuning: null argument");
    for (const Knob& k : kKnobs)
        if (!strcmp(name, k.name)) {
            PB3D_REQUIRE(value >= k.lo && value <= k.hi, "pb3d_set_tuning: %s is %s (got %d)", k.name, k.what, value);
            ctx->*k.field = value;
            return PB3D_OK;
        }
    pb3d_set_error("pb3d_set_tuning: unknown knob '%s'", name);
    return PB3D_EINVAL;
}

int pb3d_make_current(pb3d_ctx* ctx) {
    PB3D_REQUIRE(ctx != nullptr, "pb3d_make_current: null context");
    PB3D_HIP(hipSetDevice(ctx->device));
    return PB3D_OK;
}

void pb3d_destroy(pb3d_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    pb3d_comm_destroy(ctx);
    (void)pool_flush(ctx);
    for (int i = 0; i < PB3D_NSCRATCH; ++i)
        if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->stage) (void)hipHostFree(ctx->stage);
    (void)hipEventDestroy(ctx->s32_ev);
    (void)hipStreamDestroy(ctx->stream);
    free(ctx);
}

int pb3d_device_info(pb3d_ctx* ctx, char* name, int name_cap, int* compute_units, int64_t* hbm_bytes) {
    PB3D_REQUIRE(ctx != nullptr, "pb3d_device_info: null context");
    hipDeviceProp_t prop;
    PB3D_HIP(hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_cap > 0) {
        snprintf(name, (size_t)name_cap, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return PB3D_OK;
}

int pb3d_sync(pb3d_ctx* ctx) {
    PB3D_REQUIRE(ctx != nullptr, "pb3d_sync: null context");
    return pb3d_stream_sync(ctx);
}

// give every cached device block back to the driver (when a hipMalloc fails, and at destroy)
static int pool_flush(pb3d_ctx* ctx) {
    if (ctx->pool_nfree == 0) return PB3D_OK;
    PB3D_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < ctx->pool_nfree; ++i) (void)hipFree(ctx->pool_free[i].p);
    ctx->pool_nfree = 0;
    ctx->pool_cached = 0;
    return PB3D_OK;
}

// hipMalloc that empties the pool and tries once more before giving up
static hipError_t pool_malloc(pb3d_ctx* ctx, void** p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory && ctx->pool_nfree > 0) {
        (void)hipGetLastError();
        if (pool_flush(ctx) == PB3D_OK) e = hipMalloc(p, bytes);
    }
    return e;
}

int pb3d_dev_alloc(pb3d_ctx* ctx, size_t bytes, void** dptr) {
    PB3D_REQUIRE(ctx != nullptr && dptr != nullptr, "pb3d_dev_alloc: null argument");
    *dptr = nullptr;
    PB3D_HIP(hipSetDevice(ctx->device));
    // round up so that 16-byte vector tails never leave the allocation
    // (+ 64: kernels that read whole 16-byte units may run up to 15 bytes past a volume whose rows are not multiples of 16)
    size_t padded = ((bytes ? bytes : 1) + 64 + 255) & ~(size_t)255;
    // best fit among the cached blocks: at least the request, at most an eighth (+ 1 MiB) more
    int best = -1;
    for (int i = 0; i < ctx->pool_nfree; ++i) {
        const size_t b = ctx->pool_free[i].bytes;
        if (b >= padded && b <= padded + padded / 8 + (1u << 20) && (best < 0 || b < ctx->pool_free[best].bytes)) best = i;
    }
    void* p = nullptr;
    size_t got = padded;
    if (best >= 0) {
        p = ctx->pool_free[best].p; got = ctx->pool_free[best].bytes;
        ctx->pool_cached -= got;
        ctx->pool_free[best] = ctx->pool_free[--ctx->pool_nfree];
    } else {
        PB3D_HIP(pool_malloc(ctx, &p, padded));
    }
    if (ctx->pool_cap > 0) {
        if (ctx->pool_nlive < PB3D_POOL_LIVE) ctx->pool_live[ctx->pool_nlive++] = {p, got, 0};
        // (a full table only means this block will be released with hipFree instead of being kept)
    }
    *dptr = p;
    return PB3D_OK;
}

int pb3d_dev_free(pb3d_ctx* ctx, void* dptr) {
    PB3D_REQUIRE(ctx != nullptr, "pb3d_dev_free: null context");
    if (!dptr) return PB3D_OK;
    size_t bytes = 0;
    for (int i = ctx->pool_nlive - 1; i >= 0; --i)             // the most recent allocations are freed first
        if (ctx->pool_live[i].p == dptr) { bytes = ctx->pool_live[i].bytes; ctx->pool_live[i] = ctx->pool_live[--ctx->pool_nlive]; break; }
    if (bytes > 0 && bytes <= ctx->pool_cap) {
        // make room: the oldest cached blocks go back to the driver (that needs the stream idle: their last users are on it)
        while (ctx->pool_nfree > 0 && (ctx->pool_cached + bytes > ctx->pool_cap || ctx->pool_nfree == PB3D_POOL_SLOTS)) {
            int old = 0;
            for (int i = 1; i < ctx->pool_nfree; ++i)
                if (ctx->pool_free[i].stamp < ctx->pool_free[old].stamp) old = i;
            PB3D_TRY(pb3d_stream_sync(ctx));
            PB3D_HIP(hipFree(ctx->pool_free[old].p));
            ctx->pool_cached -= ctx->pool_free[old].bytes;
            ctx->pool_free[old] = ctx->pool_free[--ctx->pool_nfree];
        }
        ctx->pool_free[ctx->pool_nfree++] = {dptr, bytes, ++ctx->pool_stamp};
        ctx->pool_cached += bytes;
        return PB3D_OK;
    }
    PB3D_TRY(pb3d_stream_sync(ctx));
    PB3D_HIP(hipFree(dptr));
    return PB3D_OK;
}

int pb3d_dev_memset(pb3d_ctx* ctx, void* dptr, int value, size_t bytes) {
    PB3D_REQUIRE(ctx != nullptr && (dptr != nullptr || bytes == 0), "pb3d_dev_memset: null argument");
    if (bytes) PB3D_HIP(hipMemsetAsync(dptr, value, bytes, ctx->stream));
    return PB3D_OK;
}

int pb3d_h2d(pb3d_ctx* ctx, void* dptr, const void* hptr, size_t bytes) {
    PB3D_REQUIRE(ctx != nullptr, "pb3d_h2d: null context");
    if (!bytes) return PB3D_OK;
    PB3D_REQUIRE(dptr && hptr, "pb3d_h2d: null buffer");
    PB3D_HIP(hipMemcpyAsync(dptr, hptr, bytes, hipMemcpyHostToDevice, ctx->stream));
    return pb3d_stream_sync(ctx);
}

// Host -> device WITHOUT waiting: the bytes are copied into a pinned ring first, so the caller's buffer is free again when the call returns
// and the copy itself is ordered on the context's stream like a kernel.  Inputs larger than half the ring take the blocking path.
int pb3d_h2d_async(pb3d_ctx* ctx, void* dptr, const void* hptr, size_t bytes) {
    PB3D_REQUIRE(ctx != nullptr, "pb3d_h2d_async: null context");
    if (!bytes) return PB3D_OK;
    PB3D_REQUIRE(dptr && hptr, "pb3d_h2d_async: null buffer");
    if (!ctx->stage) {
        const size_t ring = (size_t)8 << 20;
        PB3D_HIP(hipHostMalloc(&ctx->stage, ring, hipHostMallocDefault));
        ctx->stage_bytes = ring; ctx->stage_head = 0;
    }
    if (bytes > ctx->stage_bytes / 2) return pb3d_h2d(ctx, dptr, hptr, bytes);
    size_t head = (ctx->stage_head + 63) & ~(size_t)63;
    if (head + bytes > ctx->stage_bytes) {        // the ring is full: everything staged so far must have left it
        PB3D_TRY(pb3d_stream_sync(ctx));
        head = 0;
    }
    memcpy((char*)ctx->stage + head, hptr, bytes);
    PB3D_HIP(hipMemcpyAsync(dptr, (char*)ctx->stage + head, bytes, hipMemcpyHostToDevice, ctx->stream));
    ctx->stage_head = head + bytes;
    return PB3D_OK;
}

int pb3d_d2h(pb3d_ctx* ctx, void* hptr, const void* dptr, size_t bytes) {
    PB3D_REQUIRE(ctx != nullptr, "pb3d_d2h: null context");
    if (!bytes) return PB3D_OK;
    PB3D_REQUIRE(dptr && hptr, "pb3d_d2h: null buffer");
    PB3D_HIP(hipMemcpyAsync(hptr, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return pb3d_stream_sync(ctx);
}

int pb3d_d2d(pb3d_ctx* ctx, void* dst, const void* src, size_t bytes) {
    PB3D_REQUIRE(ctx != nullptr, "pb3d_d2d: null context");
    if (!bytes) return PB3D_OK;
    PB3D_REQUIRE(dst && src, "pb3d_d2d: null buffer");
    PB3D_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return PB3D_OK;
}

int pb3d_event_create(pb3d_ctx* ctx, pb3d_event** ev) {
    PB3D_REQUIRE(ctx != nullptr && ev != nullptr, "pb3d_event_create: null argument");
    pb3d_event* e = (pb3d_event*)calloc(1, sizeof(pb3d_event));
    if (!e) {
        pb3d_set_error("pb3d_event_create: out of host memory");
        return PB3D_ENOMEM;
    }
    hipError_t he = hipEventCreate(&e->ev);
    if (he != hipSuccess) {
        free(e);
        pb3d_set_error("hipEventCreate failed: %s", hipGetErrorString(he));
        return PB3D_ENODEVICE;
    }
    *ev = e;
    return PB3D_OK;
}

int pb3d_event_record(pb3d_ctx* ctx, pb3d_event* ev) {
    PB3D_REQUIRE(ctx != nullptr && ev != nullptr, "pb3d_event_record: null argument");
    PB3D_HIP(hipEventRecord(ev->ev, ctx->stream));
    return PB3D_OK;
}

int pb3d_event_elapsed_ms(pb3d_ctx* ctx, pb3d_event* start, pb3d_event* stop, float* ms) {
    PB3D_REQUIRE(ctx && start && stop && ms, "pb3d_event_elapsed_ms: null argument");
    PB3D_HIP(hipEventSynchronize(stop->ev));
    PB3D_HIP(hipEventElapsedTime(ms, start->ev, stop->ev));
    return PB3D_OK;
}

void pb3d_event_destroy(pb3d_event* ev) {
    if (!ev) return;
    (void)hipEventDestroy(ev->ev);
    free(ev);
}

// ---- host shim H1 ------------------------------------------------------------------------------
static const uint64_t k_rotinv_bits[91][9] = {
#include "rotinv_table.inc"
};

int pb3d_rotinv(int angle_deg, double M[9]) {
    PB3D_REQUIRE(M != nullptr, "pb3d_rotinv: null output");
    PB3D_REQUIRE(angle_deg >= 0 && angle_deg <= 90, "pb3d_rotinv: angle %d outside the pinned table 0..90", angle_deg);
    memcpy(M, k_rotinv_bits[angle_deg], sizeof(double) * 9);
    return PB3D_OK;
}

int pb3d_offset(const double M[9], const int64_t shape[3], double off[3]) {
    PB3D_REQUIRE(M && shape && off, "pb3d_offset: null argument");
    // c - M@c, c = shape/2; NumPy's dgemv accumulates each row as fma(M2,c2, fma(M1,c1, M0*c0)).
    const double c[3] = {(double)shape[0] / 2.0, (double)shape[1] / 2.0, (double)shape[2] / 2.0};
    for (int h = 0; h < 3; ++h) {
        double p = M[3 * h] * c[0];
        p = std::fma(M[3 * h + 1], c[1], p);
        p = std::fma(M[3 * h + 2], c[2], p);
        off[h] = c[h] - p;
    }
    return PB3D_OK;
}

}  // extern "C"

int pb3d_stream_sync(pb3d_ctx* ctx) {
    PB3D_HIP(hipStreamSynchronize(ctx->stream));
    ctx->stage_head = 0;
    ++ctx->sync_count;
    return PB3D_OK;
}

int pb3d_scratch(pb3d_ctx* ctx, int slot, size_t bytes, void** out) {
    PB3D_REQUIRE(ctx != nullptr && slot >= 0 && slot < PB3D_NSCRATCH, "pb3d_scratch: bad slot");
    if (ctx->scratch_bytes[slot] < bytes || !ctx->scratch[slot]) {
        if (ctx->scratch[slot]) {
            PB3D_TRY(pb3d_stream_sync(ctx));
            PB3D_HIP(hipFree(ctx->scratch[slot]));
            ctx->scratch[slot] = nullptr;
            ctx->scratch_bytes[slot] = 0;
        }
        size_t padded = ((bytes ? bytes : 1) + 64 + 4095) & ~(size_t)4095;
        PB3D_HIP(pool_malloc(ctx, &ctx->scratch[slot], padded));
        ctx->scratch_bytes[slot] = padded;
        ++ctx->scratch_gen;
        ++ctx->scratch_slot_gen[slot];
    }
    *out = ctx->scratch[slot];
    return PB3D_OK;
}
