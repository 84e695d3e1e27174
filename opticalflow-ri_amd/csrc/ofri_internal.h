// Internal declarations shared by the HIP translation units of libofri_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <mutex>
#include <vector>
#include "../../include/ofri.h"

struct ofri_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    int hs_fuse = 0;                                   // sweeps per launch, 0 = automatic
    int share = 1;                                     // contexts whose launches overlap on this GPU (ofri_set_share)
    int fb_mode = 0;                                   // Farneback: 0 fast (fused multiply-adds), -1 exact (separately rounded)
    int lk_mode = 0;                                   // dense LK: 0 fast (tile kernel where it applies), -1 exact
    int ls_mode = 0;                                   // Liu-Shen: 0 automatic, 1 one sweep per launch, 6 / 8 pipeline depth, -1 exact
    int ncu = 0;                                       // compute units of the device (0 = not queried yet)
    std::map<size_t, std::vector<void *>> pool;        // free blocks by rounded size
    std::map<void *, size_t> live;                     // blocks handed out -> rounded size
    unsigned long long pool_requests = 0;              // ofri_alloc calls so far (ofri_alloc_stats)
    double *red_partials = nullptr;                    // device scratch for two-stage reductions
    size_t red_capacity = 0;
    std::map<std::string, void *> tables;              // cached device-resident constant tables
    void *pinned = nullptr;                            // 4 KiB pinned host buffer for scalar read-back
    // optional timing of the Horn-Schunck sweep launches (ofri_prof_enable)
    bool prof = false;
    struct ProfRec { hipEvent_t a, b; int rows, cols, launches, sweeps; };
    std::vector<ProfRec> prof_recs;
    std::vector<ProfRec> cap_prof_recs;                // timed sections recorded into the capture under way (they go to its graph)
    // stream capture (ofri_graph_begin .. ofri_graph_end): blocks handed out while capturing are recycled among
    // the captured work only and end up owned by the graph, so nothing else is ever given memory the graph writes
    bool capturing = false;
    std::map<size_t, std::vector<void *>> cap_pool;    // blocks freed again during the capture
    std::map<void *, size_t> cap_blocks;               // every block handed out during the capture
    // blocks handed out during a finished capture that the caller still holds (a plan's result arrays): the graph's
    // kernels write them on every replay, so when freed they join that graph's scratch memory, not the pool
    std::map<void *, struct ofri_graph *> graph_held;
};

struct ofri_graph {
    ofri_ctx *ctx;
    hipGraph_t graph;
    hipGraphExec_t exec;
    std::vector<std::pair<void *, size_t>> blocks;     // scratch memory of the captured work
    std::vector<ofri_ctx::ProfRec> prof_recs;          // timed sections inside the graph: their events are recorded at every replay
};

struct ofri_event {
    hipEvent_t ev;
    int device;
};

int ofri_set_error(int code, const char *fmt, ...);
// host waits for the context's stream; refused while the stream is being captured into a graph
int ofri_stream_wait(ofri_ctx *ctx);

#define OFRI_HIP(call)                                                                     \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess)                                                              \
            return ofri_set_error(OFRI_EHIP, "%s failed: %s (%s:%d)", #call,               \
                                  hipGetErrorString(e_), __FILE__, __LINE__);              \
    } while (0)

#define OFRI_REQUIRE(cond, ...)                                                            \
    do {                                                                                   \
        if (!(cond)) return ofri_set_error(OFRI_EINVAL, __VA_ARGS__);                      \
    } while (0)

#define OFRI_TRY(expr)                                                                     \
    do {                                                                                   \
        int r_ = (expr);                                                                   \
        if (r_ != OFRI_OK) return r_;                                                      \
    } while (0)

// kernel launch check: launch errors surface through hipGetLastError
#define OFRI_LAUNCH_CHECK()                                                                \
    do {                                                                                   \
        hipError_t e_ = hipGetLastError();                                                 \
        if (e_ != hipSuccess)                                                              \
            return ofri_set_error(OFRI_EHIP, "kernel launch failed: %s (%s:%d)",           \
                                  hipGetErrorString(e_), __FILE__, __LINE__);              \
    } while (0)

static inline int ofri_check_ctx(ofri_ctx *ctx) {
    if (!ctx) return ofri_set_error(OFRI_EINVAL, "null context");
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return ofri_set_error(OFRI_EHIP, "hipSetDevice(%d): %s", ctx->device, hipGetErrorString(e));
    return OFRI_OK;
}

// RAII block from the context pool (returned to the pool, not to the driver)
struct PoolBuf {
    ofri_ctx *ctx;
    void *p = nullptr;
    PoolBuf(ofri_ctx *c) : ctx(c) {}
    int get(size_t bytes) { return ofri_alloc(ctx, bytes, &p); }
    ~PoolBuf() { if (p) ofri_free(ctx, p); }
    template <class T> T *as() { return static_cast<T *>(p); }
};

// Device-to-device copies and fills as kernels on the context's stream.  The library never enqueues hipMemsetAsync /
// hipMemcpyAsync(DeviceToDevice) itself: inside a captured graph those become memset / memcpy nodes, which the runtime may
// hand to another engine than the kernel nodes around them, and replays of such graphs were seen to run a fill after
// the kernel that overwrote its destination (several contexts replaying on streams of one pool).  Kernel nodes of one
// stream keep their order.  bytes may be any number; 16-byte moves where both pointers allow it.
int ofri_dev_copy(ofri_ctx *ctx, void *dst, const void *src, size_t bytes);
int ofri_dev_fill(ofri_ctx *ctx, void *dst, unsigned char value, size_t bytes);
// rows of width_bytes (a multiple of 4, pointers and pitches 4-byte aligned) between pitched planes
int ofri_dev_copy2d(ofri_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch, size_t width_bytes, size_t rows);

// upload a small host table into a pooled device block (stream-ordered; the host copy is staged
// before returning because pageable hipMemcpyAsync completes its host read before it returns)
int ofri_upload_table(ofri_ctx *ctx, PoolBuf &buf, const void *host, size_t bytes);
// device copy of a host table, built once per (context, key) and kept until the context dies
int ofri_cached_table(ofri_ctx *ctx, const std::string &key, const void *host, size_t bytes, void **dptr);
static inline void *ofri_find_table(ofri_ctx *ctx, const std::string &key) {
    auto it = ctx->tables.find(key);
    return it == ctx->tables.end() ? nullptr : it->second;
}

// number of compute units of the context's device (queried once)
static inline int ofri_ctx_ncu(ofri_ctx *ctx) {
    if (ctx->ncu <= 0) {
        hipDeviceProp_t prop;
        ctx->ncu = (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0)
                       ? prop.multiProcessorCount : 256;
    }
    return ctx->ncu;
}

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// reflect-101 ("mirror", edge sample not repeated): -1 -> 1, n -> n-2; valid for |overshoot| <= n-1
__host__ __device__ static inline int refl101(int i, int n) {
    if (n == 1) return 0;
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}
// symmetric reflect (edge sample repeated): -1 -> 0, n -> n-1
__host__ __device__ static inline int reflsym(int i, int n) {
    if (i < 0) i = -i - 1;
    if (i >= n) i = 2 * n - 1 - i;
    return i;
}
__host__ __device__ static inline int clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }
