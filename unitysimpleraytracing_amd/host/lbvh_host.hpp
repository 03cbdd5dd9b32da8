// lbvh_host.hpp — the reference's C# host classes restated in C++ over the C ABI (include/lbvh.h).
//
// The reference's host side is compiled C# against UnityEngine (no C# toolchain in this image), so
// the host layer above the C ABI is C++: same class names, constructor arguments, method names and
// call order as Assets/_Scripts/{DataBuffer,MeshBufferContainer,ComputeBufferSorter,BVHConstructor,
// RaytracingMeshDrawer}.cs.  Each method is one C-ABI call; nothing here computes.  Errors: the
// reference logs and carries on; here every failing call throws lbvh::Error with the library's text.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/lbvh.h"

namespace lbvh {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string& m) : std::runtime_error("lbvh status " + std::to_string(s) + ": " + m), status(s) {}
};

inline void check(lbvh_context* ctx, lbvh_status s)
{
    if (s != LBVH_OK) throw Error(s, lbvh_last_error(ctx));
}

// The implicit Unity graphics device + IShaderContainer (Assets/_Scripts/ShaderContainer.cs:6-40).
class Context {
public:
    explicit Context(int device_id = 0)
    {
        check_abi();
        check(nullptr, lbvh_create(device_id, &ctx_));
    }
    // a context whose work is ordered by a stream the caller owns (hipStream_t)
    Context(int device_id, void* hip_stream)
    {
        check_abi();
        check(nullptr, lbvh_create_on_stream(device_id, hip_stream, &ctx_));
    }
    ~Context() { if (ctx_) lbvh_destroy(ctx_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    lbvh_context* get() const { return ctx_; }
    void sync() { check(ctx_, lbvh_sync(ctx_)); }
    // LBVH_TRACE_FAST keeps the previous frame's per-tile costs as a dispatch hint: drop it (the next frame is a first frame)
    void trace_forget() { check(ctx_, lbvh_trace_forget(ctx_)); }
    static int device_count() { return lbvh_device_count(); }
private:
    static void check_abi()
    {
        if (lbvh_abi_version() != LBVH_ABI_VERSION)
            throw Error(LBVH_ERR_INVALID_ARG, "liblbvh.so was built from another include/lbvh.h (ABI version)");
    }
    lbvh_context* ctx_ = nullptr;
};

// A HIP event on the context's stream (measurement: elapsed_ms waits for `stop`).
class Event {
public:
    explicit Event(Context& ctx) : ctx_(ctx) { check(ctx_.get(), lbvh_event_create(ctx_.get(), &ev_)); }
    ~Event() { if (ev_) lbvh_event_destroy(ctx_.get(), ev_); }
    Event(const Event&) = delete;
    Event& operator=(const Event&) = delete;
    void record() { check(ctx_.get(), lbvh_event_record(ctx_.get(), ev_)); }
    static float elapsed_ms(Event& start, Event& stop)
    {
        float ms = 0.0f;
        check(start.ctx_.get(), lbvh_event_elapsed_ms(start.ctx_.get(), start.ev_, stop.ev_, &ms));
        return ms;
    }
private:
    Context& ctx_;
    void* ev_ = nullptr;
};

// An ORDERING event (lbvh_sync_event_create): recorded on one context's stream, waited for by another's, on the device.
class SyncEvent {
public:
    explicit SyncEvent(Context& ctx) : ctx_(ctx) { check(ctx_.get(), lbvh_sync_event_create(ctx_.get(), &ev_)); }
    ~SyncEvent() { if (ev_) lbvh_event_destroy(ctx_.get(), ev_); }
    SyncEvent(const SyncEvent&) = delete;
    SyncEvent& operator=(const SyncEvent&) = delete;
    void record() { check(ctx_.get(), lbvh_event_record(ctx_.get(), ev_)); }
    // everything enqueued on `waiter` after this call starts after the work recorded before record()
    void make_wait(Context& waiter) { check(waiter.get(), lbvh_event_wait(waiter.get(), ev_)); }
private:
    Context& ctx_;
    void* ev_ = nullptr;
};

// Assets/_Scripts/DataBuffer.cs
template <typename T>
class DataBuffer {
public:
    DataBuffer(Context& ctx, size_t size) : ctx_(ctx), local_(size)                       // :25-30
    {
        check(ctx_.get(), lbvh_buffer_alloc(ctx_.get(), size, sizeof(T), &device_));
    }
    DataBuffer(Context& ctx, size_t size, uint32_t initial_word) : DataBuffer(ctx, size)  // :14-23
    {
        Fill(initial_word);
    }
    ~DataBuffer() { Dispose(); }
    DataBuffer(const DataBuffer&) = delete;
    DataBuffer& operator=(const DataBuffer&) = delete;

    void* DeviceBuffer() const { return device_; }
    std::vector<T>& LocalBuffer() { return local_; }
    size_t Size() const { return local_.size(); }

    void Fill(uint32_t word, bool mirror = true)
    {
        check(ctx_.get(), lbvh_buffer_fill_u32(ctx_.get(), device_, word, local_.size() * sizeof(T) / 4));
        if (mirror) {
            uint32_t* w = reinterpret_cast<uint32_t*>(local_.data());
            for (size_t i = 0; i < local_.size() * sizeof(T) / 4; i++) w[i] = word;
        }
    }
    void GetData()                                                                         // :50-54
    {
        check(ctx_.get(), lbvh_buffer_download(ctx_.get(), local_.data(), device_, local_.size() * sizeof(T)));
    }
    void Sync()                                                                            // :56-60
    {
        check(ctx_.get(), lbvh_buffer_upload(ctx_.get(), device_, local_.data(), local_.size() * sizeof(T)));
    }
    void Dispose()                                                                         // :72-75
    {
        if (device_) { lbvh_buffer_free(ctx_.get(), device_); device_ = nullptr; }
    }
private:
    Context& ctx_;
    void* device_ = nullptr;
    std::vector<T> local_;
};

inline uint32_t CapacityFor(uint32_t n, uint32_t tile = 1024) { return (n + tile - 1) / tile * tile; }

// Assets/_Scripts/MeshBufferContainer.cs
class MeshBufferContainer {
public:
    MeshBufferContainer(Context& ctx, const std::vector<lbvh_triangle>& triangles, uint32_t capacity = 0)
        : ctx_(ctx), n_((uint32_t)triangles.size()), cap_(capacity ? capacity : CapacityFor(n_)),
          keys_(ctx, cap_), index_(ctx, cap_), tris_(ctx, cap_), aabb_(ctx, cap_), bvh_(ctx, cap_),
          leaf_(ctx, cap_, LBVH_NULL), internal_(ctx, cap_, LBVH_NULL)                     // :108-115
    {
        std::memcpy(tris_.LocalBuffer().data(), triangles.data(), triangles.size() * sizeof(lbvh_triangle));
        tris_.Sync();                                                                      // :150
        GenerateKeys();                                                                    // :123-151 as one kernel
    }
    void GenerateKeys()
    {
        const float mn[3] = {-125.0f, -125.0f, -125.0f}, mx[3] = {125.0f, 125.0f, 125.0f}; // Whole, :9-15
        check(ctx_.get(), lbvh_morton_aabb(ctx_.get(), (const lbvh_triangle*)tris_.DeviceBuffer(), n_, cap_, mn, mx,
                                           (uint32_t*)keys_.DeviceBuffer(), (uint32_t*)index_.DeviceBuffer(),
                                           (lbvh_aabb*)aabb_.DeviceBuffer()));
    }
    void DistributeKeys()                                                                  // :154-169
    {
        check(ctx_.get(), lbvh_distribute_keys(ctx_.get(), (uint32_t*)keys_.DeviceBuffer(), n_));
    }
    void GetAllGpuData()                                                                   // :171-196
    {
        keys_.GetData(); index_.GetData(); tris_.GetData(); aabb_.GetData(); bvh_.GetData(); leaf_.GetData();
        internal_.GetData();
        for (uint32_t i = 0; i < n_; i++)
            if (leaf_.LocalBuffer()[i].index == LBVH_NULL && leaf_.LocalBuffer()[i].parent == LBVH_NULL)
                throw Error(-100, "LEAF CORRUPTED " + std::to_string(i));                  // :183-186
        for (uint32_t i = 0; i + 1 < n_; i++)
            if (internal_.LocalBuffer()[i].index == LBVH_NULL && internal_.LocalBuffer()[i].parent == LBVH_NULL)
                throw Error(-101, "INTERNAL CORRUPTED " + std::to_string(i));              // :191-194
    }
    lbvh_scene Scene() const
    {
        lbvh_scene s;
        s.n = n_;
        s.sorted_indices = (const uint32_t*)index_.DeviceBuffer();
        s.triangle_aabb = (const lbvh_aabb*)aabb_.DeviceBuffer();
        s.internal_nodes = (const lbvh_internal_node*)internal_.DeviceBuffer();
        s.leaf_nodes = (const lbvh_leaf_node*)leaf_.DeviceBuffer();
        s.bvh = (const lbvh_aabb*)bvh_.DeviceBuffer();
        s.triangles = (const lbvh_triangle*)tris_.DeviceBuffer();
        return s;
    }
    uint32_t TrianglesLength() const { return n_; }
    uint32_t Capacity() const { return cap_; }
    DataBuffer<uint32_t>& Keys() { return keys_; }
    DataBuffer<uint32_t>& TriangleIndex() { return index_; }
    DataBuffer<lbvh_triangle>& TriangleData() { return tris_; }
    DataBuffer<lbvh_aabb>& TriangleAABB() { return aabb_; }
    DataBuffer<lbvh_aabb>& BvhData() { return bvh_; }
    DataBuffer<lbvh_leaf_node>& BvhLeafNode() { return leaf_; }
    DataBuffer<lbvh_internal_node>& BvhInternalNode() { return internal_; }
private:
    Context& ctx_;
    uint32_t n_, cap_;
    DataBuffer<uint32_t> keys_, index_;
    DataBuffer<lbvh_triangle> tris_;
    DataBuffer<lbvh_aabb> aabb_, bvh_;
    DataBuffer<lbvh_leaf_node> leaf_;
    DataBuffer<lbvh_internal_node> internal_;
};

// Assets/_Scripts/ComputeBufferSorter.cs — dataLength only bounds the validation, Sort() covers the
// whole padded buffers (every dispatch of the reference covers DATA_ARRAY_COUNT, :107,116).
class ComputeBufferSorter {
public:
    ComputeBufferSorter(Context& ctx, uint32_t data_length, DataBuffer<uint32_t>& keys, DataBuffer<uint32_t>& values)
        : ctx_(ctx), data_length_(data_length), keys_(keys), values_(values) {}
    void Sort()                                                                            // :100-126
    {
        check(ctx_.get(), lbvh_sort_pairs(ctx_.get(), (uint32_t*)keys_.DeviceBuffer(), (uint32_t*)values_.DeviceBuffer(),
                                          (uint32_t)keys_.Size()));
    }
    bool ValidateSortedData()                                                              // :150-177
    {
        keys_.GetData();
        for (uint32_t i = 1; i < data_length_; i++)
            if (keys_.LocalBuffer()[i] < keys_.LocalBuffer()[i - 1]) return false;
        return true;
    }
private:
    Context& ctx_;
    uint32_t data_length_;
    DataBuffer<uint32_t>& keys_;
    DataBuffer<uint32_t>& values_;
};

// Assets/_Scripts/BVHConstructor.cs
class BVHConstructor {
public:
    BVHConstructor(Context& ctx, uint32_t triangles_count, DataBuffer<uint32_t>& sorted_morton_codes,
                   DataBuffer<uint32_t>& sorted_triangle_indices, DataBuffer<lbvh_aabb>& triangle_aabb,
                   DataBuffer<lbvh_internal_node>& internal_nodes, DataBuffer<lbvh_leaf_node>& leaf_nodes,
                   DataBuffer<lbvh_aabb>& bvh_data)
        : ctx_(ctx), n_(triangles_count), keys_(sorted_morton_codes), idx_(sorted_triangle_indices), aabb_(triangle_aabb),
          internal_(internal_nodes), leaf_(leaf_nodes), bvh_(bvh_data) {}
    void ConstructTree()                                                                   // :61-64
    {
        check(ctx_.get(), lbvh_build_tree(ctx_.get(), n_, (const uint32_t*)keys_.DeviceBuffer(),
                                          (lbvh_internal_node*)internal_.DeviceBuffer(), (lbvh_leaf_node*)leaf_.DeviceBuffer()));
    }
    void ConstructBVH()                                                                    // :66-69
    {
        check(ctx_.get(), lbvh_refit(ctx_.get(), n_, (const lbvh_internal_node*)internal_.DeviceBuffer(),
                                     (const lbvh_leaf_node*)leaf_.DeviceBuffer(), (const lbvh_aabb*)aabb_.DeviceBuffer(),
                                     (const uint32_t*)idx_.DeviceBuffer(), (lbvh_aabb*)bvh_.DeviceBuffer()));
    }
private:
    Context& ctx_;
    uint32_t n_;
    DataBuffer<uint32_t>& keys_;
    DataBuffer<uint32_t>& idx_;
    DataBuffer<lbvh_aabb>& aabb_;
    DataBuffer<lbvh_internal_node>& internal_;
    DataBuffer<lbvh_leaf_node>& leaf_;
    DataBuffer<lbvh_aabb>& bvh_;
};

// Assets/_Scripts/RaytracingMeshDrawer.cs: Awake() = build chain (:30-51), Update() = per-frame
// dispatch (:76-84) producing hit records.
class RaytracingMeshDrawer {
public:
    RaytracingMeshDrawer(Context& ctx, const std::vector<lbvh_triangle>& mesh) : ctx_(ctx), mesh_(mesh) {}
    void Awake()
    {
        container_.reset(new MeshBufferContainer(ctx_, mesh_));                                            // :34
        sorter_.reset(new ComputeBufferSorter(ctx_, container_->TrianglesLength(), container_->Keys(),
                                              container_->TriangleIndex()));                               // :36
        sorter_->Sort();                                                                                   // :37
        container_->DistributeKeys();                                                                      // :39
        bvh_.reset(new BVHConstructor(ctx_, container_->TrianglesLength(), container_->Keys(), container_->TriangleIndex(),
                                      container_->TriangleAABB(), container_->BvhInternalNode(),
                                      container_->BvhLeafNode(), container_->BvhData()));                  // :41-48
        bvh_->ConstructTree();                                                                             // :50
        bvh_->ConstructBVH();                                                                              // :51
        const lbvh_scene s = container_->Scene();
        const float mn[3] = {-125.0f, -125.0f, -125.0f}, mx[3] = {125.0f, 125.0f, 125.0f};   // MeshBufferContainer.Whole
        check(ctx_.get(), lbvh_build_fast_scene(ctx_.get(), &s, mn, mx));
    }
    // screenWidth/screenHeight/cameraFov/cameraToWorldMatrix + Dispatch, :78-83
    void Update(const lbvh_camera& cam, int mode = LBVH_TRACE_FAST)
    {
        const size_t rays = (size_t)cam.screen_width * cam.screen_height;
        if (!hits_ || hits_->Size() < rays) hits_.reset(new DataBuffer<lbvh_hit>(ctx_, rays));
        const lbvh_scene s = container_->Scene();
        check(ctx_.get(), lbvh_trace_primary(ctx_.get(), &cam, 0, 0, cam.screen_width, cam.screen_height, &s, mode,
                                             (lbvh_hit*)hits_->DeviceBuffer(), nullptr));
    }
    // one GPU's share of the frame (every shard_count-th group of 8 adjacent 8x8-pixel tiles), one launch; the hit buffer
    // has the full frame's layout and only the shard's pixels are written (BASELINE configs[2]: BVH replicated, rays sharded)
    void UpdateShard(const lbvh_camera& cam, uint32_t shard_index, uint32_t shard_count, int mode = LBVH_TRACE_FAST)
    {
        const size_t rays = (size_t)cam.screen_width * cam.screen_height;
        if (!hits_ || hits_->Size() < rays) hits_.reset(new DataBuffer<lbvh_hit>(ctx_, rays));
        const lbvh_scene s = container_->Scene();
        check(ctx_.get(), lbvh_trace_primary_shard(ctx_.get(), &cam, shard_index, shard_count, &s, mode,
                                                   (lbvh_hit*)hits_->DeviceBuffer(), nullptr));
    }
    // _objectDrawer.SetTexture("_meshTexture", ...) :61 — RGBA8, row 0 at v = 0
    void SetTexture(const std::vector<uint8_t>& rgba8, int width, int height)
    {
        tex_.reset(new DataBuffer<uint32_t>(ctx_, (size_t)width * height));
        std::memcpy(tex_->LocalBuffer().data(), rgba8.data(), (size_t)width * height * 4);
        tex_->Sync();
        tex_w_ = width; tex_h_ = height;
    }
    // the shading tail of the Raytracing kernel (Raytracing.compute:178-184) -> RGBA16F, 4 halves per pixel
    void Shade()
    {
        const size_t rays = hits_->Size();
        if (!image_ || image_->Size() < rays) image_.reset(new DataBuffer<uint64_t>(ctx_, rays));
        check(ctx_.get(), lbvh_shade(ctx_.get(), (const lbvh_hit*)hits_->DeviceBuffer(), rays,
                                     (const lbvh_triangle*)container_->TriangleData().DeviceBuffer(),
                                     (const uint8_t*)tex_->DeviceBuffer(), tex_w_, tex_h_, (uint16_t*)image_->DeviceBuffer()));
    }
    // OnRenderImage :86-90 — Graphics.Blit(src, dest, _imageComposerMaterial): the shaded image over the camera's own
    // rendering (RGBA16F, 4 halves per pixel), in place in `src_dest`
    void OnRenderImage(DataBuffer<uint64_t>& src_dest)
    {
        check(ctx_.get(), lbvh_compose(ctx_.get(), (const uint16_t*)src_dest.DeviceBuffer(), (const uint16_t*)image_->DeviceBuffer(),
                                       hits_->Size(), (uint16_t*)src_dest.DeviceBuffer()));
    }
    // per-frame rebuild on the same buffers (dynamic scenes): the whole Awake() chain in one call
    void Rebuild()
    {
        MeshBufferContainer& c = *container_;
        const float mn[3] = {-125.0f, -125.0f, -125.0f}, mx[3] = {125.0f, 125.0f, 125.0f};   // MeshBufferContainer.Whole
        check(ctx_.get(), lbvh_build_scene(ctx_.get(), (const lbvh_triangle*)c.TriangleData().DeviceBuffer(), c.TrianglesLength(),
                                           c.Capacity(), mn, mx, (uint32_t*)c.Keys().DeviceBuffer(),
                                           (uint32_t*)c.TriangleIndex().DeviceBuffer(), (lbvh_aabb*)c.TriangleAABB().DeviceBuffer(),
                                           (lbvh_internal_node*)c.BvhInternalNode().DeviceBuffer(),
                                           (lbvh_leaf_node*)c.BvhLeafNode().DeviceBuffer(), (lbvh_aabb*)c.BvhData().DeviceBuffer(),
                                           LBVH_BUILD_FAST_SCENE | LBVH_BUILD_RESET_NODES));
    }
    // this GPU's share of the frame written into `frame` — a full-frame buffer that may live on ANOTHER GPU (peer-mapped:
    // lbvh_peer_enable on this context first): the records travel as the tiles finish, nothing is copied afterwards
    void UpdateShardInto(const lbvh_camera& cam, uint32_t shard_index, uint32_t shard_count, lbvh_hit* frame, int mode = LBVH_TRACE_FAST)
    {
        const lbvh_scene s = container_->Scene();
        check(ctx_.get(), lbvh_trace_primary_shard(ctx_.get(), &cam, shard_index, shard_count, &s, mode, frame, nullptr));
    }
    DataBuffer<uint64_t>& Image() { return *image_; }
    MeshBufferContainer& Container() { return *container_; }
    DataBuffer<lbvh_hit>& Hits() { return *hits_; }
private:
    Context& ctx_;
    std::vector<lbvh_triangle> mesh_;
    std::unique_ptr<MeshBufferContainer> container_;
    std::unique_ptr<ComputeBufferSorter> sorter_;
    std::unique_ptr<BVHConstructor> bvh_;
    std::unique_ptr<DataBuffer<lbvh_hit>> hits_;
    std::unique_ptr<DataBuffer<uint32_t>> tex_;
    std::unique_ptr<DataBuffer<uint64_t>> image_;
    int tex_w_ = 0, tex_h_ = 0;
};

// One frame from N GPUs driven by ONE process — what a Unity MonoBehaviour is (Assets/_Scripts/RaytracingMeshDrawer.cs:30-84
// runs Awake / Update on the main thread of one process) and BASELINE configs[2] asks for: BVH replicated, rays sharded, no
// collective.  One Context per entry of `devices` (a device may appear more than once: logical ranks on one GPU, which is
// how the tests run it on a one-GPU box), each with its own replica of the scene; Awake() / Rebuild() enqueue every
// replica's build round-robin (all calls are asynchronous, so the N GPUs build side by side); Update() has every context
// trace its share of the frame straight into ONE full-frame hit buffer that lives on the first device — peer-mapped
// stores over xGMI while the tiles finish, no second pass — and makes the first context's stream wait for the others'
// completion events on the device.  Whatever the first context enqueues next (lbvh_shade, a download, Unity's blit) sees a
// whole frame; the host never blocks inside Update().  The next Update() may only overwrite the buffer once the owner is
// done reading the previous frame: the owner records `consumed_` at the top of Update() and every other stream waits for it.
class MultiGpuDrawer {
public:
    MultiGpuDrawer(const std::vector<int>& devices, const std::vector<lbvh_triangle>& mesh)
    {
        if (devices.empty()) throw Error(LBVH_ERR_INVALID_ARG, "MultiGpuDrawer: no devices");
        for (int d : devices) {
            contexts_.emplace_back(new Context(d));
            drawers_.emplace_back(new RaytracingMeshDrawer(*contexts_.back(), mesh));
            done_.emplace_back(new SyncEvent(*contexts_.back()));
        }
        consumed_.reset(new SyncEvent(*contexts_[0]));
        for (size_t r = 1; r < contexts_.size(); ++r)            // rank r stores into the owner's memory
            check(contexts_[r]->get(), lbvh_peer_enable(contexts_[r]->get(), devices[0]));
    }
    ~MultiGpuDrawer()
    {
        try { Sync(); } catch (const Error&) {}                            // no rank may still be storing into the frame buffer
    }
    size_t Ranks() const { return contexts_.size(); }
    Context& Owner() { return *contexts_[0]; }
    RaytracingMeshDrawer& Drawer(size_t rank) { return *drawers_[rank]; }
    void Awake() { for (auto& d : drawers_) d->Awake(); }                  // replicas: deterministic, bit-identical trees
    void Rebuild() { for (auto& d : drawers_) d->Rebuild(); }              // per-frame rebuilds of dynamic scenes
    void Update(const lbvh_camera& cam, int mode = LBVH_TRACE_FAST)
    {
        const size_t rays = (size_t)cam.screen_width * cam.screen_height;
        if (!frame_ || frame_->Size() < rays) {
            Sync();                                                        // nobody may still be writing the old buffer
            frame_.reset(new DataBuffer<lbvh_hit>(*contexts_[0], rays));
        }
        const uint32_t n = (uint32_t)contexts_.size();
        consumed_->record();                                               // the owner's reads of the previous frame end here
        for (uint32_t r = 1; r < n; ++r) consumed_->make_wait(*contexts_[r]);
        for (uint32_t r = 0; r < n; ++r) {                                 // round-robin enqueue: N GPUs trace side by side
            drawers_[r]->UpdateShardInto(cam, r, n, (lbvh_hit*)frame_->DeviceBuffer(), mode);
            if (r != 0) done_[r]->record();
        }
        for (uint32_t r = 1; r < n; ++r) done_[r]->make_wait(*contexts_[0]);   // the gather: a device-side wait, no copy
    }
    // the whole frame, on the first device; valid for work enqueued on Owner() after Update()
    DataBuffer<lbvh_hit>& Hits() { return *frame_; }
    void Sync() { for (auto& c : contexts_) c->sync(); }
private:
    // (destruction order: buffers and events before their contexts)
    std::vector<std::unique_ptr<Context>> contexts_;
    std::vector<std::unique_ptr<RaytracingMeshDrawer>> drawers_;
    std::vector<std::unique_ptr<SyncEvent>> done_;
    std::unique_ptr<SyncEvent> consumed_;
    std::unique_ptr<DataBuffer<lbvh_hit>> frame_;
};

// BASELINE configs[4] (extension): rigid bodies rotate every frame, the LBVH is rebuilt, primary rays + `bounces`
// diffuse bounces at 1 spp.  Twin of host.py DynamicPathTracer.
class DynamicPathTracer {
public:
    DynamicPathTracer(Context& ctx, const std::vector<lbvh_triangle>& rest, const std::vector<uint32_t>& body_ids,
                      const std::vector<float>& body_centres_xyzw, float t_min = 1e-3f, float albedo = 0.7f, uint32_t seed = 1)
        : ctx_(ctx), drawer_(ctx, rest), rest_(ctx, rest.size()), body_(ctx, body_ids.size()),
          centres_(ctx, body_centres_xyzw.size()), t_min_(t_min), albedo_(albedo), seed_(seed)
    {
        drawer_.Awake();
        rest_.LocalBuffer() = rest;                 rest_.Sync();
        body_.LocalBuffer() = body_ids;             body_.Sync();
        centres_.LocalBuffer() = body_centres_xyzw; centres_.Sync();
    }
    // the bodies turned to `angle` and the whole LBVH rebuilt, one call (= lbvh_animate + RaytracingMeshDrawer::Rebuild)
    void Animate(float angle)
    {
        MeshBufferContainer& c = drawer_.Container();
        const float mn[3] = {-125.0f, -125.0f, -125.0f}, mx[3] = {125.0f, 125.0f, 125.0f};   // MeshBufferContainer.Whole
        check(ctx_.get(), lbvh_animate_build_scene(ctx_.get(), (const lbvh_triangle*)rest_.DeviceBuffer(), (const uint32_t*)body_.DeviceBuffer(),
                                                   (const float*)centres_.DeviceBuffer(), std::cos(angle), std::sin(angle),
                                                   (lbvh_triangle*)c.TriangleData().DeviceBuffer(), c.TrianglesLength(), c.Capacity(), mn, mx,
                                                   (uint32_t*)c.Keys().DeviceBuffer(), (uint32_t*)c.TriangleIndex().DeviceBuffer(),
                                                   (lbvh_aabb*)c.TriangleAABB().DeviceBuffer(), (lbvh_internal_node*)c.BvhInternalNode().DeviceBuffer(),
                                                   (lbvh_leaf_node*)c.BvhLeafNode().DeviceBuffer(), (lbvh_aabb*)c.BvhData().DeviceBuffer(),
                                                   LBVH_BUILD_FAST_SCENE | LBVH_BUILD_RESET_NODES));
    }
    void Render(const lbvh_camera& cam, uint32_t bounces = 4)
    {
        const size_t rays = (size_t)cam.screen_width * cam.screen_height;
        if (!states_ || states_->Size() < rays) {
            states_.reset(new DataBuffer<lbvh_path_state>(ctx_, rays));
            image_.reset(new DataBuffer<uint64_t>(ctx_, rays));
        }
        drawer_.Update(cam);                                                     // primary rays: the packet kernel
        const lbvh_scene s = drawer_.Container().Scene();
        lbvh_path_state* st = (lbvh_path_state*)states_->DeviceBuffer();
        lbvh_hit* hits = (lbvh_hit*)drawer_.Hits().DeviceBuffer();
        // bounce b = scatter at segment b's hits + trace of segment b + 1; the first one makes the path states from the camera
        if (bounces == 0) check(ctx_.get(), lbvh_path_begin(ctx_.get(), &cam, st));
        else check(ctx_.get(), lbvh_path_first_bounce(ctx_.get(), &cam, &s, st, hits, seed_, albedo_, t_min_));
        for (uint32_t b = 1; b < bounces; ++b)
            check(ctx_.get(), lbvh_path_bounce(ctx_.get(), &s, st, hits, rays, b, seed_, albedo_, t_min_));
        check(ctx_.get(), lbvh_path_scatter(ctx_.get(), &s, hits, rays, bounces, seed_, albedo_, st));
        check(ctx_.get(), lbvh_path_resolve(ctx_.get(), st, rays, (uint16_t*)image_->DeviceBuffer()));
    }
    DataBuffer<uint64_t>& Image() { return *image_; }
    RaytracingMeshDrawer& Drawer() { return drawer_; }
private:
    Context& ctx_;
    RaytracingMeshDrawer drawer_;
    DataBuffer<lbvh_triangle> rest_;
    DataBuffer<uint32_t> body_;
    DataBuffer<float> centres_;
    std::unique_ptr<DataBuffer<lbvh_path_state>> states_;
    std::unique_ptr<DataBuffer<uint64_t>> image_;
    float t_min_, albedo_;
    uint32_t seed_;
};

}  // namespace lbvh
