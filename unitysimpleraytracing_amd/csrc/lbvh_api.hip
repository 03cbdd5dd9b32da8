// lbvh_api.hip — context, buffers, events: the DataBuffer/ComputeBuffer half of the C ABI
// (reference: Assets/_Scripts/DataBuffer.cs, Assets/_Scripts/ShaderContainer.cs).
#include <algorithm>
#include <cstdlib>
#include "lbvh_common.h"

#include <string.h>

static thread_local std::string g_create_error;

int lbvh_set_error(lbvh_context* ctx, int code, const char* what, const char* detail)
{
    std::string msg = std::string(what ? what : "") + (detail ? std::string(": ") + detail : "");
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}

int lbvh_reserve(lbvh_context* ctx, void** ptr, size_t* have, size_t bytes)
{
    if (*have >= bytes && *ptr) return LBVH_OK;
    if (*ptr) {
        // earlier launches may still be using the old block
        LBVH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->side_stream) LBVH_HIP_TRY(ctx, hipStreamSynchronize(ctx->side_stream));
        LBVH_HIP_TRY(ctx, hipFree(*ptr));
        *ptr = nullptr;
        *have = 0;
    }
    // (from here to the end a failure leaves the slot EMPTY — null pointer, zero bytes: the next call of whatever owns it starts over)
    *ptr = nullptr;
    *have = 0;
    size_t want = bytes < 256 ? 256 : bytes;
    if (ctx->debug_switch[LBVH_DEBUG_FAIL_RESERVE] != 0 && --ctx->debug_switch[LBVH_DEBUG_FAIL_RESERVE] == 0)      // (test hook)
        return lbvh_set_error(ctx, LBVH_ERR_OUT_OF_MEMORY, "lbvh_reserve", "hipMalloc failure injected by LBVH_DEBUG_FAIL_RESERVE");
    void* fresh = nullptr;
    const hipError_t e = hipMalloc(&fresh, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();              // (the runtime's sticky "last error": this one is reported through the status)
        return lbvh_set_error(ctx, e == hipErrorOutOfMemory ? LBVH_ERR_OUT_OF_MEMORY : LBVH_ERR_HIP, "hipMalloc(scratch)", hipGetErrorString(e));
    }
    *ptr = fresh;
    *have = want;
    return LBVH_OK;
}

int lbvh_ensure_side(lbvh_context* ctx)
{
    if (!ctx->side_stream) {
        LBVH_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
        LBVH_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        LBVH_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
        LBVH_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_hier, hipEventDisableTiming));
    }
    return LBVH_OK;
}

static bool ranges_overlap(const void* a, size_t a_bytes, const void* b, size_t b_bytes)
{
    const uintptr_t a0 = (uintptr_t)a, b0 = (uintptr_t)b;
    return a && b && a_bytes && b_bytes && a0 < b0 + b_bytes && b0 < a0 + a_bytes;
}

void lbvh_note_write(lbvh_context* ctx, const void* p, size_t bytes)
{
    // path states / hit records written from outside: the live-path list of the last bounce no longer describes them
    if (ctx->ray_list.valid && (ranges_overlap(p, bytes, ctx->ray_list.states, ctx->ray_list.count * sizeof(lbvh_path_state)) ||
                                ranges_overlap(p, bytes, ctx->ray_list.hits, ctx->ray_list.count * sizeof(lbvh_hit))))
        ctx->ray_list.valid = false;
    if (!ctx->fast_valid) return;
    const size_t n = ctx->fast_src.n;
    if (ranges_overlap(p, bytes, ctx->fast_src.triangles, n * sizeof(lbvh_triangle)) ||
        ranges_overlap(p, bytes, ctx->fast_src.sorted_indices, n * sizeof(uint32_t)) ||
        ranges_overlap(p, bytes, ctx->fast_src.triangle_aabb, n * sizeof(lbvh_aabb)))
        ctx->fast_valid = false;
}

void lbvh_note_fast_built(lbvh_context* ctx, const lbvh_scene& s)
{
    ctx->fast_src.triangles = s.triangles;
    ctx->fast_src.sorted_indices = s.sorted_indices;
    ctx->fast_src.triangle_aabb = s.triangle_aabb;
    ctx->fast_src.n = s.n;
    ctx->fast_valid = true;
    ctx->wide_valid = false;      // the four-wide form is of the previous tree
}

int lbvh_require_fast(lbvh_context* ctx, const lbvh_scene& s, const char* who)
{
    if (!ctx->fast_nodes || ctx->fast_src.n != s.n || ctx->fast_src.triangles != s.triangles ||
        ctx->fast_src.sorted_indices != s.sorted_indices || ctx->fast_src.triangle_aabb != s.triangle_aabb)
        return lbvh_set_error(ctx, LBVH_ERR_INVALID_ARG, who, "needs lbvh_build_fast_scene on this scene first");
    if (!ctx->fast_valid)
        return lbvh_set_error(ctx, LBVH_ERR_INVALID_ARG, who,
                              "the derived traversal scene is stale: the scene's triangles / sorted indices / triangle AABBs "
                              "were written after lbvh_build_fast_scene — build it again");
    return LBVH_OK;
}

int lbvh_check_fault(lbvh_context* ctx)
{
    if (!ctx->fault_host) return LBVH_OK;
    // reported ONCE: the word is taken (exchanged with 0), so the sync / download that follows the failed work returns
    // LBVH_ERR_HIP and later work on the context (a repeated sort, the next rebuild) is judged on its own (the
    // word used to stay set for the life of the context).  Callers run after a stream sync: no kernel is writing it.
    const uint32_t code = __atomic_exchange_n(ctx->fault_host, 0u, __ATOMIC_RELAXED);
    if (code == 0) return LBVH_OK;
    char msg[128];
    snprintf(msg, sizeof msg, "%s (fault %u): results enqueued before this call are invalid",
             code == LBVH_FAULT_RAY_STACK    ? "a per-ray traversal stack ran out of entries"
             : code == LBVH_FAULT_FRAME_WAIT ? "lbvh_frame_wait: a rank's completion flag never arrived"
                                             : "a bounded inter-workgroup wait gave up", code);
    return lbvh_set_error(ctx, LBVH_ERR_HIP, "device-side protocol fault", msg);
}

hipEvent_t lbvh_prof_event(lbvh_context* ctx)
{
    hipEvent_t e = nullptr;
    if (!ctx->prof_pool.empty()) {
        e = ctx->prof_pool.back();
        ctx->prof_pool.pop_back();
        return e;
    }
    (void)hipEventCreateWithFlags(&e, hipEventDisableSystemFence);
    return e;
}

extern "C" {

int32_t lbvh_abi_version(void) { return LBVH_ABI_VERSION; }

int32_t lbvh_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

static lbvh_status create_impl(int32_t device_id, void* stream, bool own, lbvh_context** out_ctx)
{
    if (!out_ctx) return lbvh_set_error(nullptr, LBVH_ERR_INVALID_ARG, "lbvh_create", "out_ctx is null");
    *out_ctx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return lbvh_set_error(nullptr, LBVH_ERR_NO_DEVICE, "lbvh_create", "no HIP device visible");
    if (device_id < 0 || device_id >= n)
        return lbvh_set_error(nullptr, LBVH_ERR_INVALID_ARG, "lbvh_create", "device_id out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess)
        return lbvh_set_error(nullptr, LBVH_ERR_HIP, "lbvh_create", "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0 || prop.warpSize != 64)
        return lbvh_set_error(nullptr, LBVH_ERR_NO_DEVICE, "lbvh_create: device is not gfx950/wave64",
                              prop.gcnArchName);
    lbvh_context* ctx = new lbvh_context();
    ctx->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess) {
        delete ctx;
        return lbvh_set_error(nullptr, LBVH_ERR_HIP, "lbvh_create", "hipSetDevice failed");
    }
    if (own) {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return lbvh_set_error(nullptr, LBVH_ERR_HIP, "lbvh_create", "hipStreamCreate failed");
        }
        ctx->own_stream = true;
    } else {
        ctx->stream = (hipStream_t)stream;
        ctx->own_stream = false;
    }
    ctx->cur_stream = ctx->stream;
    // 8 per-XCD ticket queues for the sort only on the layout they assume (in CPX/DPX/QPX partitions or
    // behind a CU-masked stream the workgroups of a launch do not visit all eight XCDs in turn): the whole 256-CU
    // device and a stream whose CU mask (if it can be read at all) enables every CU
    {
        bool full = prop.multiProcessorCount == 256;
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (full && hipExtStreamGetCUMask(ctx->stream, 8, mask) == hipSuccess) {
            for (int i = 0; i < 8; i++) full = full && mask[i] == 0xFFFFFFFFu;
        } else if (!own) {
            full = false;           // a caller's stream whose mask cannot be read: assume nothing
        }
        (void)hipGetLastError();
        ctx->sort_queues = ctx->sort_queues_detected = full ? 8u : 1u;      // (tests force a mode: lbvh_debug_switch)
    }
    if (hipHostMalloc((void**)&ctx->fault_host, 256, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&ctx->fault_dev, ctx->fault_host, 0) != hipSuccess) {
        if (ctx->fault_host) (void)hipHostFree(ctx->fault_host);
        if (own) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return lbvh_set_error(nullptr, LBVH_ERR_OUT_OF_MEMORY, "lbvh_create", "no mapped host memory for the fault word");
    }
    memset(ctx->fault_host, 0, 256);
    // words 16 .. 19: the largest (balanced) bucket of the last sort, tagged with its fine shift (lbvh_sort.hip, the two-level form's hint); nothing known yet
    for (int i = 16; i < 20; i++) ctx->fault_host[i] = 0xFFFFFFFFu;
    ctx->fault_host[24] = 0xFFFFFFFFu;        // word 24: the work estimate of the frame before last (lbvh_trace.hip, launch_packets); nothing known yet
    *out_ctx = ctx;
    return LBVH_OK;
}

lbvh_status lbvh_create(int32_t device_id, lbvh_context** out_ctx)
{
    return create_impl(device_id, nullptr, true, out_ctx);
}

lbvh_status lbvh_create_on_stream(int32_t device_id, void* hip_stream, lbvh_context** out_ctx)
{
    return create_impl(device_id, hip_stream, false, out_ctx);
}

lbvh_status lbvh_destroy(lbvh_context* ctx)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->sort_scratch) (void)hipFree(ctx->sort_scratch);
    if (ctx->side_stream) (void)hipStreamSynchronize(ctx->side_stream);
    for (int l = 0; l < 2; l++) {
        if (ctx->scan_scratch[l]) (void)hipFree(ctx->scan_scratch[l]);
        if (ctx->refit_scratch[l]) (void)hipFree(ctx->refit_scratch[l]);
    }
    if (ctx->build_graph) (void)hipGraphExecDestroy(ctx->build_graph);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->ev_hier) (void)hipEventDestroy(ctx->ev_hier);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->fast_nodes) (void)hipFree(ctx->fast_nodes);
    if (ctx->trace_queues) (void)hipFree(ctx->trace_queues);
    if (ctx->fast_tree) (void)hipFree(ctx->fast_tree);
    if (ctx->hier) (void)hipFree(ctx->hier);
    if (ctx->ray_scratch) (void)hipFree(ctx->ray_scratch);
    if (ctx->tie_list) (void)hipFree(ctx->tie_list);
    if (ctx->wide_nodes) (void)hipFree(ctx->wide_nodes);
    if (ctx->trace_frame_costs) (void)hipFree(ctx->trace_frame_costs);
    for (auto& s : ctx->prof_spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    for (auto e : ctx->prof_pool) (void)hipEventDestroy(e);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->fault_host) (void)hipHostFree(ctx->fault_host);
    delete ctx;
    return LBVH_OK;
}

const char* lbvh_last_error(const lbvh_context* ctx)
{
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

lbvh_status lbvh_sync(lbvh_context* ctx)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return lbvh_check_fault(ctx);
}

// ---- buffers -------------------------------------------------------------------------------

lbvh_status lbvh_buffer_alloc(lbvh_context* ctx, size_t count, size_t stride, void** out_d_ptr)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, out_d_ptr != nullptr);
    LBVH_REQUIRE(ctx, stride > 0);
    *out_d_ptr = nullptr;
    size_t bytes = count * stride;
    if (bytes == 0) bytes = stride;
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    // (count * stride must not wrap: 2^44 elements of 2^20 bytes is not a small buffer)
    if (count != 0 && bytes / count != stride)
        return lbvh_set_error(ctx, LBVH_ERR_OUT_OF_MEMORY, "lbvh_buffer_alloc", "count * stride overflows size_t");
    const hipError_t e = hipMalloc(out_d_ptr, bytes);
    if (e != hipSuccess) {
        *out_d_ptr = nullptr;
        (void)hipGetLastError();
        return lbvh_set_error(ctx, e == hipErrorOutOfMemory ? LBVH_ERR_OUT_OF_MEMORY : LBVH_ERR_HIP, "hipMalloc(buffer)", hipGetErrorString(e));
    }
    return LBVH_OK;
}

lbvh_status lbvh_buffer_free(lbvh_context* ctx, void* d_ptr)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    if (!d_ptr) return LBVH_OK;
    LBVH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (d_ptr == ctx->ray_list.states || d_ptr == ctx->ray_list.hits) ctx->ray_list.valid = false;
    // a freed scene buffer can no longer back the derived scene (its address may be handed out again)
    if (ctx->fast_valid && (d_ptr == ctx->fast_src.triangles || d_ptr == ctx->fast_src.sorted_indices ||
                            d_ptr == ctx->fast_src.triangle_aabb)) {
        ctx->fast_valid = false;
        ctx->fast_src.n = 0;
    }
    LBVH_HIP_TRY(ctx, hipFree(d_ptr));
    return LBVH_OK;
}

lbvh_status lbvh_buffer_fill_u32(lbvh_context* ctx, void* d_ptr, uint32_t value, size_t n_words)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    if (n_words == 0) return LBVH_OK;
    LBVH_REQUIRE(ctx, d_ptr != nullptr);
    lbvh_note_write(ctx, d_ptr, n_words * 4);
    LBVH_HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)d_ptr, (int)value, n_words, ctx->stream));
    return LBVH_OK;
}

lbvh_status lbvh_buffer_upload(lbvh_context* ctx, void* d_dst, const void* h_src, size_t bytes)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    if (bytes == 0) return LBVH_OK;
    LBVH_REQUIRE(ctx, d_dst != nullptr && h_src != nullptr);
    lbvh_note_write(ctx, d_dst, bytes);
    LBVH_HIP_TRY(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    // pageable host memory: the caller may reuse h_src as soon as we return
    LBVH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return LBVH_OK;
}

lbvh_status lbvh_buffer_download(lbvh_context* ctx, void* h_dst, const void* d_src, size_t bytes)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    if (bytes == 0) return LBVH_OK;
    LBVH_REQUIRE(ctx, h_dst != nullptr && d_src != nullptr);
    LBVH_HIP_TRY(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    LBVH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return lbvh_check_fault(ctx);
}

// ---- events ----------------------------------------------------------------------------------

lbvh_status lbvh_event_create(lbvh_context* ctx, void** out_event)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, out_event != nullptr);
    // timing-only events: no system-scope fence at the record (HIP: "avoids the cost of cache writeback and
    // invalidation, and the performance impact of those actions on the execution of following work")
    hipEvent_t ev;
    LBVH_HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableSystemFence));
    *out_event = (void*)ev;
    return LBVH_OK;
}

lbvh_status lbvh_event_destroy(lbvh_context* ctx, void* event)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    if (!event) return LBVH_OK;
    LBVH_HIP_TRY(ctx, hipEventDestroy((hipEvent_t)event));
    return LBVH_OK;
}

lbvh_status lbvh_event_record(lbvh_context* ctx, void* event)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, event != nullptr);
    LBVH_HIP_TRY(ctx, hipEventRecord((hipEvent_t)event, ctx->stream));
    return LBVH_OK;
}

lbvh_status lbvh_event_elapsed_ms(lbvh_context* ctx, void* start, void* stop, float* out_ms)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, start != nullptr && stop != nullptr && out_ms != nullptr);
    LBVH_HIP_TRY(ctx, hipEventSynchronize((hipEvent_t)stop));
    LBVH_HIP_TRY(ctx, hipEventElapsedTime(out_ms, (hipEvent_t)start, (hipEvent_t)stop));
    return LBVH_OK;
}

// ---- one frame from N GPUs: peer access, cross-context / cross-process completion ----------------------------------------

lbvh_status lbvh_peer_enable(lbvh_context* ctx, int32_t peer_device)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    int n = 0;
    LBVH_HIP_TRY(ctx, hipGetDeviceCount(&n));
    LBVH_REQUIRE(ctx, peer_device >= 0 && peer_device < n);
    if (peer_device == ctx->device) return LBVH_OK;
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int can = 0;
    LBVH_HIP_TRY(ctx, hipDeviceCanAccessPeer(&can, ctx->device, peer_device));
    if (!can) return lbvh_set_error(ctx, LBVH_ERR_HIP, "lbvh_peer_enable", "the two GPUs cannot access each other's memory");
    const hipError_t e = hipDeviceEnablePeerAccess(peer_device, 0);
    if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return LBVH_OK; }
    LBVH_HIP_TRY(ctx, e);
    return LBVH_OK;
}

lbvh_status lbvh_sync_event_create(lbvh_context* ctx, void** out_event)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, out_event != nullptr);
    // an ORDERING event: its record is a system-scope release (what the timing events of lbvh_event_create leave out), so
    // whoever waits for it — another GPU's stream included — sees every store enqueued before it
    // hipEventReleaseToSystem spelled out: whether a plain event's record releases to system scope is a runtime default
    // (HIP_EVENT_SYS_RELEASE), and the documented contract must not depend on it
    hipEvent_t ev;
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    LBVH_HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventReleaseToSystem));
    *out_event = (void*)ev;
    return LBVH_OK;
}

lbvh_status lbvh_event_wait(lbvh_context* ctx, void* event)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, event != nullptr);
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    LBVH_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, (hipEvent_t)event, 0));
    return LBVH_OK;
}

static_assert(sizeof(hipIpcMemHandle_t) == LBVH_IPC_HANDLE_BYTES, "lbvh.h promises 64-byte IPC handles");

lbvh_status lbvh_ipc_export(lbvh_context* ctx, void* d_ptr, uint8_t h_handle[LBVH_IPC_HANDLE_BYTES])
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, d_ptr != nullptr && h_handle != nullptr);
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipIpcMemHandle_t h;
    LBVH_HIP_TRY(ctx, hipIpcGetMemHandle(&h, d_ptr));
    memcpy(h_handle, &h, sizeof h);
    return LBVH_OK;
}

lbvh_status lbvh_ipc_import(lbvh_context* ctx, const uint8_t h_handle[LBVH_IPC_HANDLE_BYTES], void** out_d_ptr)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, h_handle != nullptr && out_d_ptr != nullptr);
    *out_d_ptr = nullptr;
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipIpcMemHandle_t h;
    memcpy(&h, h_handle, sizeof h);
    LBVH_HIP_TRY(ctx, hipIpcOpenMemHandle(out_d_ptr, h, hipIpcMemLazyEnablePeerAccess));
    return LBVH_OK;
}

lbvh_status lbvh_ipc_close(lbvh_context* ctx, void* d_ptr)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    if (!d_ptr) return LBVH_OK;
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    LBVH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    LBVH_HIP_TRY(ctx, hipIpcCloseMemHandle(d_ptr));
    return LBVH_OK;
}

// Completion flags between GPUs that share no process (bench.py: one rank per GPU).  The signalling kernel runs after the
// stream's earlier kernels have ended (their stores have left this GPU's caches: the end-of-kernel release) and publishes
// with a system-scope release store; flag and data take the same xGMI path to the owner's memory.  The waiting kernel polls
// with system-scope acquire loads — its own HBM written by the peers, or the owner's word read over xGMI — and ends when every
// flag has reached the frame number; what the stream runs after it starts with the usual kernel-start invalidate and reads the peers' records.
__global__ void frame_signal_kernel(uint32_t* flag, uint32_t value)
{
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The bound is WALL CLOCK (s_memrealtime: the constant 100 MHz counter), not a poll count: the waiting side may be enqueued
// long before the signalling process has even initialised its GPU, and how fast polls come back depends on where the
// flag lives (own HBM, or the owner's over xGMI).
__global__ __launch_bounds__(64) void frame_wait_kernel(const uint32_t* flags, uint32_t n_slots, uint32_t value, uint32_t* fault,
                                                        unsigned long long timeout_ticks)
{
    const uint32_t lane = threadIdx.x;
    bool here = lane >= n_slots;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        if (!here) here = (int32_t)(__hip_atomic_load(flags + lane, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - value) >= 0;
        if (__builtin_amdgcn_ballot_w64(!here) == 0) return;
        if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) break;
        __builtin_amdgcn_s_sleep(16);
    }
    if (lane == 0) __hip_atomic_store(fault, LBVH_FAULT_FRAME_WAIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

lbvh_status lbvh_flags_alloc(lbvh_context* ctx, size_t n_words, uint32_t** out_d_flags)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, out_d_flags != nullptr && n_words > 0);
    *out_d_flags = nullptr;
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    void* p = nullptr;
    // uncached: a kernel that polls these words while another GPU / process stores into them must read the memory, not a
    // line its own L2 may keep for as long as the kernel runs (coarse-grained hipMalloc memory promises nothing before the
    // kernel's end) — ADVICE r4.  256-byte granules so that no two flag arrays ever share a line.
    const size_t bytes = (n_words * 4 + 255) & ~(size_t)255;
    LBVH_HIP_TRY(ctx, hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached));
    const hipError_t e = hipMemsetAsync(p, 0, bytes, ctx->stream);
    if (e != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
        (void)hipFree(p);
        return lbvh_set_error(ctx, LBVH_ERR_HIP, "lbvh_flags_alloc", "clearing the flag words failed");
    }
    *out_d_flags = (uint32_t*)p;
    return LBVH_OK;
}

lbvh_status lbvh_debug_switch(lbvh_context* ctx, uint32_t which, uint32_t value)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, which < LBVH_DEBUG_SWITCHES);
    switch (which) {
    case LBVH_DEBUG_SORT_QUEUES:
        LBVH_REQUIRE(ctx, value == 0 || value == 1 || value == 8);
        ctx->sort_queues = value ? value : ctx->sort_queues_detected;
        break;
    case LBVH_DEBUG_COLD_ORDER: LBVH_REQUIRE(ctx, value <= 1); break;
    case LBVH_DEBUG_BUILD_FORM:
        LBVH_REQUIRE(ctx, value <= 4);
        // a captured graph belongs to the form it was captured under
        if (value != ctx->debug_switch[which] && ctx->build_graph) {
            LBVH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            (void)hipGraphExecDestroy(ctx->build_graph);
            ctx->build_graph = nullptr;
            ctx->build_graph_key = 0;
        }
        ctx->build_graph_off = false;
        break;
    case LBVH_DEBUG_SORT_FORM: LBVH_REQUIRE(ctx, value <= 2); break;
    default: break;
    }
    ctx->debug_switch[which] = value;
    return LBVH_OK;
}

lbvh_status lbvh_frame_signal(lbvh_context* ctx, uint32_t* d_flags, uint32_t slot, uint32_t value)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, d_flags != nullptr && ((uintptr_t)d_flags & 3) == 0);
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    LBVH_LAUNCH(ctx, frame_signal_kernel, dim3(1), dim3(1), d_flags + slot, value);
    LBVH_HIP_TRY(ctx, hipGetLastError());
    return LBVH_OK;
}

lbvh_status lbvh_frame_wait(lbvh_context* ctx, const uint32_t* d_flags, uint32_t n_slots, uint32_t value)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, n_slots <= 64u);
    if (n_slots == 0) return LBVH_OK;
    LBVH_REQUIRE(ctx, d_flags != nullptr && ((uintptr_t)d_flags & 3) == 0);
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t ms = ctx->debug_switch[LBVH_DEBUG_FRAME_WAIT_MS] ? ctx->debug_switch[LBVH_DEBUG_FRAME_WAIT_MS] : 20000u;
    LBVH_LAUNCH(ctx, frame_wait_kernel, dim3(1), dim3(64), d_flags, n_slots, value, ctx->fault_dev, (unsigned long long)ms * 100000ull);
    LBVH_HIP_TRY(ctx, hipGetLastError());
    return LBVH_OK;
}

// ---- per-kernel profiling --------------------------------------------------------------------

lbvh_status lbvh_profile_begin(lbvh_context* ctx)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    for (auto& s : ctx->prof_spans) { ctx->prof_pool.push_back(s.a); ctx->prof_pool.push_back(s.b); }
    ctx->prof_spans.clear();
    ctx->prof_enabled = true;
    return LBVH_OK;
}

lbvh_status lbvh_profile_end(lbvh_context* ctx, lbvh_profile_row* h_rows, int32_t max_rows, int32_t* out_rows)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, out_rows != nullptr && (h_rows != nullptr || max_rows == 0));
    ctx->prof_enabled = false;
    LBVH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    int32_t n = 0;
    for (auto& s : ctx->prof_spans) {
        float ms = 0.0f;
        LBVH_HIP_TRY(ctx, hipEventElapsedTime(&ms, s.a, s.b));
        // launch sites spell templated kernels as "(name<...>)": drop the parentheses
        char nm[sizeof(h_rows[0].name)];
        {
            const char* src = s.name;
            size_t len = strlen(src);
            if (len >= 2 && src[0] == '(' && src[len - 1] == ')') { src++; len -= 2; }
            if (len > sizeof(nm) - 1) len = sizeof(nm) - 1;
            memcpy(nm, src, len);
            nm[len] = 0;
        }
        int32_t row = -1;
        for (int32_t i = 0; i < n; i++)
            if (strncmp(h_rows[i].name, nm, sizeof(h_rows[i].name) - 1) == 0) { row = i; break; }
        if (row < 0 && n < max_rows) {
            row = n++;
            memset(&h_rows[row], 0, sizeof(h_rows[row]));
            strncpy(h_rows[row].name, nm, sizeof(h_rows[row].name) - 1);
        }
        if (row >= 0) { h_rows[row].launches++; h_rows[row].total_ms += ms; }
        ctx->prof_pool.push_back(s.a);
        ctx->prof_pool.push_back(s.b);
    }
    ctx->prof_spans.clear();
    *out_rows = n;
    return LBVH_OK;
}

// ---- shader-clock probe ------------------------------------------------------------------------

__global__ __launch_bounds__(256) void clock_probe_kernel(uint2* __restrict__ out, uint32_t iterations, float seed)
{
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    float a = seed + (float)threadIdx.x, b = seed * 0.5f, c = 1.0f, d = 0.25f;
    for (uint32_t i = 0; i < iterations; i++) {          // four dependent chains: the vector ALUs stay busy
#pragma unroll
        for (int k = 0; k < 4; k++) {
            a = a * 0.999f + b;
            b = b * 1.001f - c;
            c = c * 0.998f + d;
            d = d * 1.002f - a;
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (a + b + c + d == 12345.678f) out[0] = make_uint2(0u, 0u);          // keeps the loop alive
    if ((threadIdx.x & 63u) == 0)
        out[blockIdx.x * 4u + (threadIdx.x >> 6)] = make_uint2((uint32_t)(c1 - c0), (uint32_t)(r1 - r0));
}

lbvh_status lbvh_clock_probe(lbvh_context* ctx, float* out_shader_mhz)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, out_shader_mhz != nullptr);
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t blocks = 2048, waves = blocks * 4;
    uint2* d_out = nullptr;
    LBVH_HIP_TRY(ctx, hipMalloc((void**)&d_out, waves * sizeof(uint2)));
    std::vector<uint2> h(waves);
    std::vector<double> ratio;
    // the second launch is the measured one: the first brings the clock up
    for (int rep = 0; rep < 2; rep++)
        LBVH_LAUNCH(ctx, clock_probe_kernel, dim3(blocks), dim3(256), d_out, 6000u, 1.0f + (float)rep);
    hipError_t e = hipGetLastError();                     // a failed launch must not be read back as a clock
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_out, waves * sizeof(uint2), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_out);
    LBVH_HIP_TRY(ctx, e);
    for (const uint2& w : h)
        if (w.y > 0) ratio.push_back((double)w.x / (double)w.y);
    LBVH_REQUIRE(ctx, !ratio.empty());
    std::nth_element(ratio.begin(), ratio.begin() + ratio.size() / 2, ratio.end());
    *out_shader_mhz = (float)(ratio[ratio.size() / 2] * 100.0);          // s_memrealtime ticks at 100 MHz
    return LBVH_OK;
}

// ---- HBM copy-rate probe ---------------------------------------------------------------------

// one float4 per thread, no loop: the shape that reaches this chip's best copy rate (6.27 TB/s measured;
// grid-stride loops with 4-8 loads in flight per lane, block-contiguous chunks, nontemporal hints and
// hipMemcpyDtoD all land between 4.6 and 5.6 TB/s — tools/ubench/copybw.hip)
__global__ __launch_bounds__(256) void copy_f4_kernel(float4* __restrict__ dst,
                                                      const float4* __restrict__ src, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

lbvh_status lbvh_copy_bandwidth_probe(lbvh_context* ctx, void* d_dst, const void* d_src, size_t bytes)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, d_dst != nullptr && d_src != nullptr);
    LBVH_REQUIRE(ctx, bytes % 16 == 0);
    const size_t n16 = bytes / 16;
    if (n16 == 0) return LBVH_OK;
    const size_t blocks = (n16 + 255) / 256;
    LBVH_REQUIRE(ctx, blocks <= 0x7FFFFFFFu);
    LBVH_LAUNCH(ctx, copy_f4_kernel, dim3((unsigned)blocks), dim3(256), (float4*)d_dst, (const float4*)d_src, n16);
    LBVH_HIP_TRY(ctx, hipGetLastError());
    return LBVH_OK;
}

}  // extern "C"
