// Context, pooled device memory, copies, events, small conversions and reductions.
#include "ofri_internal.h"
#include <atomic>
#include <mutex>

static thread_local char g_err[512] = "";

int ofri_set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int ofri_version(void) { return 100; }

int ofri_stream_wait(ofri_ctx *ctx) {
    if (ctx->capturing)
        return ofri_set_error(OFRI_EINVAL, "this call waits for the GPU, which a captured sequence cannot do (ofri_graph_begin is active)");
    OFRI_HIP(hipStreamSynchronize(ctx->stream));
    return OFRI_OK;
}
extern "C" const char *ofri_last_error(void) { return g_err; }

extern "C" int ofri_device_count(int *count) {
    OFRI_REQUIRE(count, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return ofri_set_error(OFRI_ENODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return OFRI_OK;
}

int ofri_hs_static_setup(int device);

// Every allocating runtime call this library makes (device and pinned memory, events) is counted; ofri_alloc_stats reads it.
std::atomic<unsigned long long> g_ofri_runtime_allocs{0};
#define OFRI_COUNT_ALLOC() (void)g_ofri_runtime_allocs.fetch_add(1, std::memory_order_relaxed)

extern "C" int ofri_ctx_create(int device, void *stream, ofri_ctx **out) {
    OFRI_REQUIRE(out, "out is null");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return ofri_set_error(OFRI_ENODEVICE, "no HIP device available (libofri_hip needs an AMD GPU; there is no CPU fallback)");
    OFRI_REQUIRE(device >= 0 && device < n, "device %d out of range (0..%d)", device, n - 1);
    OFRI_HIP(hipSetDevice(device));
    // Address of the allocation-free entry point's scratch array on this GPU.  A failure here does not stop the context:
    // ofri_hs_compute_static reports the missing address when it is called, every other entry point works without it.
    if (ofri_hs_static_setup(device) != OFRI_OK) (void)hipGetLastError();
    ofri_ctx *c = new ofri_ctx();
    c->device = device;
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return ofri_set_error(OFRI_EHIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        c->owns_stream = true;
    }
    OFRI_COUNT_ALLOC();
    hipError_t e = hipHostMalloc(&c->pinned, 4096, hipHostMallocDefault);
    if (e != hipSuccess) { delete c; return ofri_set_error(OFRI_EHIP, "hipHostMalloc: %s", hipGetErrorString(e)); }
    *out = c;
    return OFRI_OK;
}

// ---- guard bands (debugging aid): with OFRI_POOL_GUARD=<KiB> in the environment every pool block is allocated with that many
// KiB of 0xC3 bytes before and after it, and ofri_pool_guard_check reports blocks whose bands were written to (a kernel that
// stores outside its arrays).  Off by default: nothing below changes the normal path.
static size_t guard_bytes() {
    static const size_t g = [] {
        const char *e = getenv("OFRI_POOL_GUARD");
        return e ? (size_t)strtoul(e, nullptr, 10) * 1024 : (size_t)0;
    }();
    return g;
}
static int guard_fill() {            // OFRI_POOL_GUARD_NAN=1: bands of 0xFF bytes (float NaNs), so that reads outside an array reach the results
    static const int v = getenv("OFRI_POOL_GUARD_NAN") ? 0xFF : 0xC3;
    return v;
}
static std::map<void *, size_t> g_guarded;          // every guarded block of the process (user pointer -> size)
static std::mutex g_guarded_mu;
// OFRI_POOL_POISON=1 (debugging aid): every block handed out is first filled with 0xFF bytes (float NaNs) in the order of the
// context's stream, so that a kernel reading memory nobody wrote in this call shows up as NaNs in the results.
static bool pool_poison() {
    static const bool on = getenv("OFRI_POOL_POISON") != nullptr;
    return on;
}
static hipError_t pool_malloc(void **p, size_t rb) {
    const size_t G = guard_bytes();
    if (!G) return hipMalloc(p, rb);
    char *base = nullptr;
    hipError_t e = hipMalloc((void **)&base, rb + 2 * G);
    if (e != hipSuccess) return e;
    (void)hipMemset(base, guard_fill(), G);
    (void)hipMemset(base + G + rb, guard_fill(), G);
    *p = base + G;
    std::lock_guard<std::mutex> lock(g_guarded_mu);
    g_guarded[*p] = rb;
    return hipSuccess;
}
static void pool_free(void *p) {
    if (guard_bytes()) {
        std::lock_guard<std::mutex> lock(g_guarded_mu);
        g_guarded.erase(p);
    }
    (void)hipFree(guard_bytes() ? (void *)((char *)p - guard_bytes()) : p);
}

extern "C" int ofri_pool_guard_check(ofri_ctx *ctx, int *bad_blocks, unsigned long long *first_bad_size, long long *first_bad_offset) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(bad_blocks, "null argument");
    *bad_blocks = 0;
    const size_t G = guard_bytes();
    OFRI_REQUIRE(G, "guard bands are off (set OFRI_POOL_GUARD=<KiB> before the library is loaded)");
    OFRI_HIP(hipDeviceSynchronize());
    std::vector<unsigned char> h(G);
    std::vector<std::pair<void *, size_t>> blocks;         // of every context: a stray store may come from any of them
    {
        std::lock_guard<std::mutex> lock(g_guarded_mu);
        for (auto &kv : g_guarded) blocks.push_back({kv.first, kv.second});
    }
    for (auto &b : blocks)
        for (int side = 0; side < 2; ++side) {
            const char *src = side ? (const char *)b.first + b.second : (const char *)b.first - G;
            OFRI_HIP(hipMemcpy(h.data(), src, G, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < G; ++i)
                if (h[i] != guard_fill()) {
                    if (!*bad_blocks) {
                        if (first_bad_size) *first_bad_size = b.second;
                        if (first_bad_offset) *first_bad_offset = side ? (long long)(b.second + i) : (long long)i - (long long)G;
                    }
                    fprintf(stderr, "ofri guard: block %p of %zu bytes: byte at offset %lld was overwritten (0x%02x)\n", b.first, b.second,
                            side ? (long long)(b.second + i) : (long long)i - (long long)G, h[i]);
                    *bad_blocks += 1;
                    break;
                }
        }
    return OFRI_OK;
}

extern "C" int ofri_pool_trim(ofri_ctx *ctx) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_TRY(ofri_stream_wait(ctx));
    for (auto &kv : ctx->pool)
        for (void *p : kv.second) pool_free(p);
    ctx->pool.clear();
    return OFRI_OK;
}

extern "C" int ofri_ctx_destroy(ofri_ctx *ctx) {
    if (!ctx) return OFRI_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->pool)
        for (void *p : kv.second) pool_free(p);
    for (auto &kv : ctx->live) pool_free(kv.first);
    for (auto &r : ctx->prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto &kv : ctx->tables) (void)hipFree(kv.second);
    if (ctx->red_partials) (void)hipFree(ctx->red_partials);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return OFRI_OK;
}

extern "C" int ofri_ctx_sync(ofri_ctx *ctx) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_TRY(ofri_stream_wait(ctx));
    return OFRI_OK;
}

extern "C" int ofri_ctx_stream(ofri_ctx *ctx, void **stream) {
    OFRI_REQUIRE(ctx && stream, "null argument");
    *stream = (void *)ctx->stream;
    return OFRI_OK;
}

static size_t round_block(size_t bytes) {
    const size_t g = bytes <= (1u << 20) ? 4096 : (size_t)(2u << 20);   // small: 4 KiB, large: 2 MiB granules
    return (bytes + g - 1) / g * g;
}

extern "C" int ofri_alloc(ofri_ctx *ctx, size_t bytes, void **dptr) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(dptr, "dptr is null");
    *dptr = nullptr;
    OFRI_REQUIRE(bytes > 0, "zero-byte allocation");
    ctx->pool_requests += 1;
    size_t rb = round_block(bytes);
    void *p = nullptr;
    if (ctx->capturing) {
        auto ci = ctx->cap_pool.find(rb);
        if (ci != ctx->cap_pool.end() && !ci->second.empty()) {
            p = ci->second.back();
            ci->second.pop_back();
            ctx->live[p] = rb;
            *dptr = p;
            if (pool_poison()) OFRI_TRY(ofri_dev_fill(ctx, p, 0xFF, rb));
            return OFRI_OK;
        }
    }
    auto it = ctx->pool.find(rb);
    if (it != ctx->pool.end() && !it->second.empty()) {
        p = it->second.back();
        it->second.pop_back();
    } else {
        OFRI_COUNT_ALLOC();
        hipError_t e = pool_malloc(&p, rb);
        if (e != hipSuccess) {
            // release cached blocks and retry once
            (void)ofri_pool_trim(ctx);
            e = pool_malloc(&p, rb);
            if (e != hipSuccess) return ofri_set_error(OFRI_ENOMEM, "hipMalloc(%zu): %s", rb, hipGetErrorString(e));
        }
    }
    ctx->live[p] = rb;
    if (ctx->capturing) ctx->cap_blocks[p] = rb;
    *dptr = p;
    if (pool_poison()) OFRI_TRY(ofri_dev_fill(ctx, p, 0xFF, rb));
    return OFRI_OK;
}

extern "C" int ofri_alloc_stats(ofri_ctx *ctx, unsigned long long *runtime_allocs, unsigned long long *pool_requests) {
    OFRI_REQUIRE(ctx, "null context");
    if (runtime_allocs) *runtime_allocs = g_ofri_runtime_allocs.load(std::memory_order_relaxed);
    if (pool_requests) *pool_requests = ctx->pool_requests;
    return OFRI_OK;
}

extern "C" int ofri_free(ofri_ctx *ctx, void *dptr) {
    if (!dptr) return OFRI_OK;
    OFRI_REQUIRE(ctx, "null context");
    auto it = ctx->live.find(dptr);
    OFRI_REQUIRE(it != ctx->live.end(), "pointer %p was not allocated by this context", dptr);
    // stream-ordered reuse: every consumer of the pool enqueues on ctx->stream, so a block handed
    // out again is only touched after the work that last used it
    auto gh = ctx->graph_held.find(dptr);
    if (ctx->capturing && ctx->cap_blocks.count(dptr)) ctx->cap_pool[it->second].push_back(dptr);
    else if (gh != ctx->graph_held.end()) {           // a captured graph still writes this block: it stays the graph's
        gh->second->blocks.push_back({dptr, it->second});
        ctx->graph_held.erase(gh);
    } else {
        if (ctx->capturing && pool_poison())
            fprintf(stderr, "ofri pool: block %p (%zu bytes), handed out before the capture, is freed during it\n", dptr, it->second);
        ctx->pool[it->second].push_back(dptr);
    }
    ctx->live.erase(it);
    return OFRI_OK;
}

// ---- captured launch sequences ---------------------------------------------------------------------------
extern "C" int ofri_graph_begin(ofri_ctx *ctx) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(!ctx->capturing, "a capture is already running on this context");
    OFRI_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
    ctx->capturing = true;
    ctx->cap_prof_recs.clear();
    ctx->cap_pool.clear();
    ctx->cap_blocks.clear();
    return OFRI_OK;
}

extern "C" int ofri_graph_end(ofri_ctx *ctx, ofri_graph **out) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(out, "out is null");
    *out = nullptr;
    OFRI_REQUIRE(ctx->capturing, "no capture is running on this context");
    ctx->capturing = false;
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(ctx->stream, &g);
    // the blocks recycled inside the capture belong to the graph (or go back to the pool if there is none)
    std::vector<std::pair<void *, size_t>> owned;
    for (auto &kv : ctx->cap_pool)
        for (void *p : kv.second) owned.push_back({p, kv.first});
    ctx->cap_pool.clear();
    std::vector<void *> held;                            // handed out during the capture and not freed yet
    for (auto &kv : ctx->cap_blocks)
        if (ctx->live.count(kv.first)) held.push_back(kv.first);
    ctx->cap_blocks.clear();
    hipGraphExec_t exec = nullptr;
    hipError_t e2 = hipSuccess;
    if (e == hipSuccess && g) e2 = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    std::vector<ofri_ctx::ProfRec> timed = std::move(ctx->cap_prof_recs);
    ctx->cap_prof_recs.clear();
    if (e != hipSuccess || !g || e2 != hipSuccess) {
        if (g) (void)hipGraphDestroy(g);
        for (auto &r : timed) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        for (auto &b : owned) ctx->pool[b.second].push_back(b.first);
        (void)hipGetLastError();
        return ofri_set_error(OFRI_EHIP, "stream capture failed: %s (a call that waits for the GPU or copies to pageable "
                              "host memory cannot be captured)", hipGetErrorString(e != hipSuccess ? e : e2));
    }
    ofri_graph *gr = new ofri_graph();
    gr->ctx = ctx; gr->graph = g; gr->exec = exec; gr->blocks = std::move(owned); gr->prof_recs = std::move(timed);
    for (void *p : held) ctx->graph_held[p] = gr;
    *out = gr;
    return OFRI_OK;
}

extern "C" int ofri_graph_launch(ofri_ctx *ctx, ofri_graph *graph) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(graph && graph->ctx == ctx, "the graph was captured on another context");
    OFRI_REQUIRE(!ctx->capturing, "a capture is running on this context");
    if (pool_poison()) {            // debugging aid: the graph's own blocks must be nobody else's
        for (auto &b : graph->blocks) {
            if (ctx->live.count(b.first)) fprintf(stderr, "ofri pool: graph block %p (%zu bytes) is also handed out\n", b.first, b.second);
            for (auto &kv : ctx->pool)
                for (void *p : kv.second)
                    if (p == b.first) fprintf(stderr, "ofri pool: graph block %p (%zu bytes) is also in the free list\n", b.first, b.second);
        }
    }
    OFRI_HIP(hipGraphLaunch(graph->exec, ctx->stream));
    return OFRI_OK;
}

// Timed sections captured into the graph (the capture ran with ofri_prof_enable on): their count, and one section's
// duration in the graph's last replay (waits for that replay's end of the section).
extern "C" int ofri_graph_prof_count(ofri_graph *graph, int *count) {
    OFRI_REQUIRE(graph && count, "null argument");
    *count = (int)graph->prof_recs.size();
    return OFRI_OK;
}
extern "C" int ofri_graph_prof_read(ofri_graph *graph, int idx, float *ms, int *launches, int *sweeps, int *rows, int *cols) {
    OFRI_REQUIRE(graph && ms && launches && sweeps && rows && cols, "null argument");
    OFRI_REQUIRE(idx >= 0 && idx < (int)graph->prof_recs.size(), "record index out of range");
    auto &r = graph->prof_recs[idx];
    OFRI_HIP(hipEventSynchronize(r.b));
    OFRI_HIP(hipEventElapsedTime(ms, r.a, r.b));
    *launches = r.launches; *sweeps = r.sweeps; *rows = r.rows; *cols = r.cols;
    return OFRI_OK;
}

// the context must still exist; the graph's scratch memory goes back to its pool
extern "C" int ofri_graph_destroy(ofri_graph *graph) {
    if (!graph) return OFRI_OK;
    (void)hipGraphExecDestroy(graph->exec);
    (void)hipGraphDestroy(graph->graph);
    for (auto &r : graph->prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto &b : graph->blocks) graph->ctx->pool[b.second].push_back(b.first);
    for (auto it = graph->ctx->graph_held.begin(); it != graph->ctx->graph_held.end();)      // what the caller still holds is plain memory again
        it = it->second == graph ? graph->ctx->graph_held.erase(it) : std::next(it);
    delete graph;
    return OFRI_OK;
}

extern "C" int ofri_upload(ofri_ctx *ctx, void *dst, const void *src_host, size_t bytes) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(dst && src_host, "null pointer");
    if (!bytes) return OFRI_OK;
    OFRI_HIP(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    OFRI_TRY(ofri_stream_wait(ctx));
    return OFRI_OK;
}

extern "C" int ofri_download(ofri_ctx *ctx, void *dst_host, const void *src, size_t bytes) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(dst_host && src, "null pointer");
    if (!bytes) return OFRI_OK;
    OFRI_HIP(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    OFRI_TRY(ofri_stream_wait(ctx));
    return OFRI_OK;
}

// ---- copies and fills as kernels (see ofri_internal.h) ------------------------------------------------------
__global__ void __launch_bounds__(256) k_dev_copy16(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16,
                                                    unsigned char *__restrict__ dtail, const unsigned char *__restrict__ stail, int ntail) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t st = (size_t)gridDim.x * 256;
    for (; i + 3 * st < n16; i += 4 * st) {          // four 16-byte moves in flight per thread
        const uint4 a = src[i], b = src[i + st], c = src[i + 2 * st], d = src[i + 3 * st];
        dst[i] = a; dst[i + st] = b; dst[i + 2 * st] = c; dst[i + 3 * st] = d;
    }
    for (; i < n16; i += st) dst[i] = src[i];
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) dtail[threadIdx.x] = stail[threadIdx.x];
}
__global__ void __launch_bounds__(256) k_dev_copy1(unsigned char *__restrict__ dst, const unsigned char *__restrict__ src, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (const size_t st = (size_t)gridDim.x * 256; i < n; i += st) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) k_dev_copy4(unsigned *__restrict__ dst, const unsigned *__restrict__ src, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (const size_t st = (size_t)gridDim.x * 256; i < n; i += st) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) k_dev_fill16(uint4 *__restrict__ dst, size_t n16, unsigned v, unsigned char *__restrict__ dtail, int ntail) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const uint4 q = make_uint4(v, v, v, v);
    for (const size_t st = (size_t)gridDim.x * 256; i < n16; i += st) dst[i] = q;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) dtail[threadIdx.x] = (unsigned char)v;
}
__global__ void __launch_bounds__(256) k_dev_fill1(unsigned char *__restrict__ dst, size_t n, unsigned char v) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (const size_t st = (size_t)gridDim.x * 256; i < n; i += st) dst[i] = v;
}
// one workgroup row per plane row: 4-byte elements (planes whose widths or pitches are not multiples of 16 bytes)
__global__ void __launch_bounds__(256) k_dev_copy2d(unsigned *__restrict__ dst, size_t dpitch4, const unsigned *__restrict__ src, size_t spitch4,
                                                    int width4, int rows) {
    for (int y = blockIdx.y; y < rows; y += gridDim.y)
        for (int x = blockIdx.x * 256 + threadIdx.x; x < width4; x += gridDim.x * 256) dst[(size_t)y * dpitch4 + x] = src[(size_t)y * spitch4 + x];
}

static int copy_grid(size_t units) {
    size_t g = (units + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

int ofri_dev_copy(ofri_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!bytes || dst == src) return OFRI_OK;
    if (((((size_t)dst) | ((size_t)src)) & 15) == 0) {
        const size_t n16 = bytes / 16;
        const int tail = (int)(bytes - 16 * n16);
        k_dev_copy16<<<copy_grid((n16 + 3) / 4), 256, 0, ctx->stream>>>((uint4 *)dst, (const uint4 *)src, n16, (unsigned char *)dst + 16 * n16,
                                                                         (const unsigned char *)src + 16 * n16, tail);
    } else if (((((size_t)dst) | ((size_t)src) | bytes) & 3) == 0) {
        k_dev_copy4<<<copy_grid(bytes / 4), 256, 0, ctx->stream>>>((unsigned *)dst, (const unsigned *)src, bytes / 4);
    } else {
        k_dev_copy1<<<copy_grid(bytes), 256, 0, ctx->stream>>>((unsigned char *)dst, (const unsigned char *)src, bytes);
    }
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}

int ofri_dev_fill(ofri_ctx *ctx, void *dst, unsigned char value, size_t bytes) {
    if (!bytes) return OFRI_OK;
    if ((((size_t)dst) & 15) == 0) {
        const size_t n16 = bytes / 16;
        const int tail = (int)(bytes - 16 * n16);
        k_dev_fill16<<<copy_grid(n16), 256, 0, ctx->stream>>>((uint4 *)dst, n16, 0x01010101u * value, (unsigned char *)dst + 16 * n16, tail);
    } else {
        k_dev_fill1<<<copy_grid(bytes), 256, 0, ctx->stream>>>((unsigned char *)dst, bytes, value);
    }
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}

int ofri_dev_copy2d(ofri_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch, size_t width_bytes, size_t rows) {
    if (!width_bytes || !rows) return OFRI_OK;
    OFRI_REQUIRE(((((size_t)dst) | ((size_t)src) | dpitch | spitch | width_bytes) & 3) == 0, "pitched copy of rows that are not whole 4-byte words");
    OFRI_REQUIRE(rows <= 0x7fffffff && width_bytes / 4 <= 0x7fffffff, "pitched copy too large");
    const int w4 = (int)(width_bytes / 4);
    dim3 g((unsigned)((w4 + 255) / 256 > 64 ? 64 : (w4 + 255) / 256), (unsigned)(rows > 4096 ? 4096 : rows));
    k_dev_copy2d<<<g, 256, 0, ctx->stream>>>((unsigned *)dst, dpitch / 4, (const unsigned *)src, spitch / 4, w4, (int)rows);
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}

extern "C" int ofri_copy(ofri_ctx *ctx, void *dst, const void *src, size_t bytes) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(dst && src, "null pointer");
    return ofri_dev_copy(ctx, dst, src, bytes);
}

extern "C" int ofri_zero(ofri_ctx *ctx, void *dst, size_t bytes) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(dst, "null pointer");
    return ofri_dev_fill(ctx, dst, 0, bytes);
}

int ofri_upload_table(ofri_ctx *ctx, PoolBuf &buf, const void *host, size_t bytes) {
    OFRI_REQUIRE(!ctx->capturing, "per-call host tables cannot be part of a captured sequence");
    OFRI_TRY(buf.get(bytes));
    OFRI_HIP(hipMemcpyAsync(buf.p, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return OFRI_OK;
}

int ofri_cached_table(ofri_ctx *ctx, const std::string &key, const void *host, size_t bytes, void **dptr) {
    void *p = ofri_find_table(ctx, key);
    if (!p) {
        OFRI_REQUIRE(!ctx->capturing, "a constant table is missing: run the sequence once before capturing it");
        OFRI_COUNT_ALLOC();
        OFRI_HIP(hipMalloc(&p, bytes));
        hipError_t e = hipMemcpy(p, host, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(p); return ofri_set_error(OFRI_EHIP, "table upload: %s", hipGetErrorString(e)); }
        ctx->tables[key] = p;
    }
    *dptr = p;
    return OFRI_OK;
}

// ---- events ------------------------------------------------------------------------------------
extern "C" int ofri_event_create(ofri_ctx *ctx, ofri_event **ev) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(ev, "ev is null");
    ofri_event *e = new ofri_event();
    e->device = ctx->device;
    OFRI_COUNT_ALLOC();
    hipError_t r = hipEventCreate(&e->ev);
    if (r != hipSuccess) { delete e; return ofri_set_error(OFRI_EHIP, "hipEventCreate: %s", hipGetErrorString(r)); }
    *ev = e;
    return OFRI_OK;
}
extern "C" int ofri_event_record(ofri_ctx *ctx, ofri_event *ev) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(ev, "ev is null");
    OFRI_HIP(hipEventRecord(ev->ev, ctx->stream));
    return OFRI_OK;
}
extern "C" int ofri_event_elapsed_ms(ofri_event *start, ofri_event *stop, float *ms) {
    OFRI_REQUIRE(start && stop && ms, "null argument");
    OFRI_HIP(hipSetDevice(stop->device));
    OFRI_HIP(hipEventSynchronize(stop->ev));
    OFRI_HIP(hipEventElapsedTime(ms, start->ev, stop->ev));
    return OFRI_OK;
}
extern "C" int ofri_event_sync(ofri_event *ev) {
    OFRI_REQUIRE(ev, "ev is null");
    OFRI_HIP(hipSetDevice(ev->device));
    OFRI_HIP(hipEventSynchronize(ev->ev));
    return OFRI_OK;
}
extern "C" int ofri_event_destroy(ofri_event *ev) {
    if (!ev) return OFRI_OK;
    (void)hipSetDevice(ev->device);
    (void)hipEventDestroy(ev->ev);
    delete ev;
    return OFRI_OK;
}

// ---- element-wise helpers ------------------------------------------------------------------------
template <class T>
__global__ void k_to_f32(const T *__restrict__ s, float *__restrict__ d, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    if ((((size_t)s) & (4 * sizeof(T) - 1)) == 0 && (((size_t)d) & 15) == 0) {       // four samples per load, 16-byte stores
        typedef T T4 __attribute__((ext_vector_type(4)));
        const T4 *s4 = reinterpret_cast<const T4 *>(s);
        float4 *d4 = reinterpret_cast<float4 *>(d);
        const size_t n4 = n / 4;
        for (; i < n4; i += st) { const T4 v = s4[i]; d4[i] = make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w); }
        i = 4 * n4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    }
    for (; i < n; i += st) d[i] = (float)s[i];
}
static int grid_for(size_t n, int block, int per_thread = 1) {        // per_thread: elements a thread takes per round
    size_t g = (n / per_thread + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}
extern "C" int ofri_u8_to_f32(ofri_ctx *ctx, const unsigned char *src, float *dst, size_t n) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(src && dst, "null pointer");
    if (!n) return OFRI_OK;
    k_to_f32<unsigned char><<<grid_for(n, 256, 4), 256, 0, ctx->stream>>>(src, dst, n);
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}
extern "C" int ofri_u16_to_f32(ofri_ctx *ctx, const unsigned short *src, float *dst, size_t n) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(src && dst, "null pointer");
    if (!n) return OFRI_OK;
    k_to_f32<unsigned short><<<grid_for(n, 256, 4), 256, 0, ctx->stream>>>(src, dst, n);
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}

// (x / 65535 * 255) in float32, applied to both frames only when the first frame's maximum exceeds 255
// (the 16-bit -> 8-bit range rule of the reference's benchmark script); the decision is taken on the GPU
__global__ void k_range_rule(float *__restrict__ a, float *__restrict__ b, size_t n, const double *__restrict__ maxa) {
    if (!((float)maxa[0] > 255.0f)) return;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) {
        a[i] = __fmul_rn(__fdiv_rn(a[i], 65535.0f), 255.0f);
        b[i] = __fmul_rn(__fdiv_rn(b[i], 65535.0f), 255.0f);
    }
}

__global__ void k_add_inplace(float *__restrict__ acc, const float *__restrict__ x, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    if (((((size_t)acc) | ((size_t)x)) & 15) == 0) {
        float4 *a4 = reinterpret_cast<float4 *>(acc);
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
        const size_t n4 = n / 4;
        for (; i < n4; i += st) {
            const float4 p = a4[i], q = x4[i];
            a4[i] = make_float4(__fadd_rn(p.x, q.x), __fadd_rn(p.y, q.y), __fadd_rn(p.z, q.z), __fadd_rn(p.w, q.w));
        }
        i = 4 * n4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    }
    for (; i < n; i += st) acc[i] = acc[i] + x[i];
}
__global__ void k_scale_inplace(float *__restrict__ y, float s, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    if ((((size_t)y) & 15) == 0) {
        float4 *y4 = reinterpret_cast<float4 *>(y);
        const size_t n4 = n / 4;
        for (; i < n4; i += st) { const float4 p = y4[i]; y4[i] = make_float4(__fmul_rn(p.x, s), __fmul_rn(p.y, s), __fmul_rn(p.z, s), __fmul_rn(p.w, s)); }
        i = 4 * n4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    }
    for (; i < n; i += st) y[i] = y[i] * s;
}
extern "C" int ofri_add_inplace(ofri_ctx *ctx, float *acc, const float *x, size_t n) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(acc && x, "null pointer");
    if (!n) return OFRI_OK;
    k_add_inplace<<<grid_for(n, 256, 4), 256, 0, ctx->stream>>>(acc, x, n);
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}
extern "C" int ofri_scale_inplace(ofri_ctx *ctx, float *y, float s, size_t n) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(y, "null pointer");
    if (!n) return OFRI_OK;
    k_scale_inplace<<<grid_for(n, 256, 4), 256, 0, ctx->stream>>>(y, s, n);
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}

// ---- reductions ----------------------------------------------------------------------------------
// Two-stage, fixed-shape tree (no atomics) so results are reproducible run to run.
#define RED_BLOCKS 1024
__device__ inline double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ inline float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
    return v;
}
__global__ void k_sqdiff_partial(const float *__restrict__ a, const float *__restrict__ b, size_t n, double *__restrict__ part) {
    __shared__ double sm[4];
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    double s = 0.0;
    for (; i < n; i += st) {
        double d = b ? (double)a[i] - (double)b[i] : (double)a[i];
        s += d * d;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}
__global__ void k_sum_final(const double *__restrict__ part, int n, double *__restrict__ out, int take_sqrt) {
    __shared__ double sm[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += part[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = sm[0] + sm[1] + sm[2] + sm[3];
        out[0] = take_sqrt ? sqrt(t) : t;
    }
}
__global__ void k_max_partial(const float *__restrict__ a, size_t n, double *__restrict__ part) {
    __shared__ float sm[4];
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    float m = -INFINITY;
    for (; i < n; i += st) m = fmaxf(m, a[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (double)fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}
__global__ void k_max_final(const double *__restrict__ part, int n, double *__restrict__ out) {
    __shared__ float sm[4];
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, (float)part[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (double)fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

static int ensure_red(ofri_ctx *ctx) {
    if (!ctx->red_partials) {
        OFRI_COUNT_ALLOC();
        OFRI_HIP(hipMalloc((void **)&ctx->red_partials, (RED_BLOCKS + 64) * sizeof(double)));
        ctx->red_capacity = RED_BLOCKS + 64;
    }
    return OFRI_OK;
}

double *ofri_red_slot(ofri_ctx *ctx, int slot);
// device-side |a-b|_F into *out through the caller's RED_BLOCKS partial sums (both in device memory); no host sync
int ofri_red_blocks(void) { return RED_BLOCKS; }
int ofri_norm_diff_at(ofri_ctx *ctx, const float *a, const float *b, size_t n, double *partials, double *out) {
    int g = grid_for(n, 256);
    if (g > RED_BLOCKS) g = RED_BLOCKS;
    k_sqdiff_partial<<<g, 256, 0, ctx->stream>>>(a, b, n, partials);
    OFRI_LAUNCH_CHECK();
    k_sum_final<<<1, 256, 0, ctx->stream>>>(partials, g, out, 1);
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}
// the same into slot `slot` (0..63) of the context's result area
int ofri_norm_diff_dev(ofri_ctx *ctx, const float *a, const float *b, size_t n, int slot) {
    OFRI_TRY(ensure_red(ctx));
    return ofri_norm_diff_at(ctx, a, b, n, ctx->red_partials, ctx->red_partials + RED_BLOCKS + slot);
}
int ofri_max_dev(ofri_ctx *ctx, const float *a, size_t n, int slot) {
    OFRI_TRY(ensure_red(ctx));
    int g = grid_for(n, 256);
    if (g > RED_BLOCKS) g = RED_BLOCKS;
    k_max_partial<<<g, 256, 0, ctx->stream>>>(a, n, ctx->red_partials);
    OFRI_LAUNCH_CHECK();
    k_max_final<<<1, 256, 0, ctx->stream>>>(ctx->red_partials, g, ctx->red_partials + RED_BLOCKS + slot);
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}
double *ofri_red_slot(ofri_ctx *ctx, int slot) { return ctx->red_partials + RED_BLOCKS + slot; }
// count (<= 512) doubles from device memory to the host through the context's pinned buffer; waits for the stream
int ofri_read_doubles(ofri_ctx *ctx, const double *dev, int count, double *out_host) {
    OFRI_HIP(hipMemcpyAsync(ctx->pinned, dev, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    OFRI_TRY(ofri_stream_wait(ctx));
    memcpy(out_host, ctx->pinned, count * sizeof(double));
    return OFRI_OK;
}
int ofri_read_slots(ofri_ctx *ctx, int first, int count, double *out_host) {
    return ofri_read_doubles(ctx, ofri_red_slot(ctx, first), count, out_host);
}

extern "C" int ofri_norm_diff(ofri_ctx *ctx, const float *a, const float *b, size_t n, double *out_host) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(a && out_host, "null pointer");
    if (!n) { *out_host = 0.0; return OFRI_OK; }
    OFRI_TRY(ofri_norm_diff_dev(ctx, a, b, n, 0));
    return ofri_read_slots(ctx, 0, 1, out_host);
}
extern "C" int ofri_max(ofri_ctx *ctx, const float *a, size_t n, float *out_host) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(a && out_host && n > 0, "null pointer or empty array");
    OFRI_TRY(ofri_max_dev(ctx, a, n, 0));
    double d;
    OFRI_TRY(ofri_read_slots(ctx, 0, 1, &d));
    *out_host = (float)d;
    return OFRI_OK;
}

int ofri_max_dev(ofri_ctx *ctx, const float *a, size_t n, int slot);
extern "C" int ofri_range_rule_16bit(ofri_ctx *ctx, float *frame0, float *frame1, size_t n) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(frame0 && frame1, "null pointer");
    if (!n) return OFRI_OK;
    OFRI_TRY(ofri_max_dev(ctx, frame0, n, 4));
    k_range_rule<<<grid_for(n, 256), 256, 0, ctx->stream>>>(frame0, frame1, n, ofri_red_slot(ctx, 4));
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}

// pinned host staging and stream-ordered uploads, for overlapping the next pair's H2D copy with compute
extern "C" int ofri_host_alloc(size_t bytes, void **hptr) {
    OFRI_REQUIRE(hptr && bytes > 0, "bad argument");
    OFRI_COUNT_ALLOC();
    OFRI_HIP(hipHostMalloc(hptr, bytes, hipHostMallocDefault));
    return OFRI_OK;
}
extern "C" int ofri_host_free(void *hptr) {
    if (hptr) OFRI_HIP(hipHostFree(hptr));
    return OFRI_OK;
}
extern "C" int ofri_upload_async(ofri_ctx *ctx, void *dst, const void *src_pinned_host, size_t bytes) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(dst && src_pinned_host, "null pointer");
    if (!bytes) return OFRI_OK;
    OFRI_HIP(hipMemcpyAsync(dst, src_pinned_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return OFRI_OK;
}
extern "C" int ofri_download_async(ofri_ctx *ctx, void *dst_pinned_host, const void *src, size_t bytes) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(dst_pinned_host && src, "null pointer");
    if (!bytes) return OFRI_OK;
    OFRI_HIP(hipMemcpyAsync(dst_pinned_host, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return OFRI_OK;
}
// make this context's stream wait for everything enqueued so far on another context's stream
extern "C" int ofri_ctx_wait(ofri_ctx *ctx, ofri_ctx *other) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(other && other->device == ctx->device, "contexts must share a device");
    hipEvent_t ev;
    OFRI_COUNT_ALLOC();
    OFRI_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    OFRI_HIP(hipEventRecord(ev, other->stream));
    OFRI_HIP(hipStreamWaitEvent(ctx->stream, ev, 0));
    OFRI_HIP(hipEventDestroy(ev));
    return OFRI_OK;
}
