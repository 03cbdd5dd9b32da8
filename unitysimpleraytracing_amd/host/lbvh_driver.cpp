// lbvh_driver.cpp — BASELINE config 1 ("plumbing") through the C++ host classes: 4 096 random
// triangles, 256x256 primary rays, the reference's Awake() + Update() call order.  Prints stage
// timings and checksums the parity tests compare with the oracle.  Links only liblbvh.so.
//     lbvh_driver [n] [w] [h] [dynamic]          seeded random triangles (cfg1)
//     lbvh_driver obj <file.obj> [w] [h] [z]     mesh ingest (lbvh_mesh.hpp) -> Awake -> Update, camera at (0, 0, z)
//     lbvh_driver multi <ranks> [n] [w] [h]      one process, <ranks> contexts (MultiGpuDrawer; rank r on device r % device
//                                                count): frames assembled in rank 0's buffer by peer-mapped stores,
//                                                compared word for word with one context's whole frame
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "lbvh_host.hpp"
#include "lbvh_mesh.hpp"

static uint64_t splitmix64(uint64_t& s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static float uniform(uint64_t& s, float lo, float hi) { return lo + (hi - lo) * (float)((splitmix64(s) >> 40) * (1.0 / 16777216.0)); }

// doubled: every second triangle is a copy of the one before it — every hit on one is an exact t tie with the other
static std::vector<lbvh_triangle> random_mesh(uint32_t n, bool doubled = false)
{
    std::vector<lbvh_triangle> mesh(n);
    uint64_t seed = 1;
    for (auto& t : mesh) {
        if (doubled && ((&t - mesh.data()) & 1)) { t = *(&t - 1); continue; }
        std::memset(&t, 0, sizeof t);
        for (int k = 0; k < 3; k++) {
            const float c = uniform(seed, -100.0f, 100.0f);
            t.a[k] = c;
            t.b[k] = c + uniform(seed, -2.0f, 2.0f);
            t.c[k] = c + uniform(seed, -2.0f, 2.0f);
        }
        t.b_uv[0] = 1.0f; t.c_uv[1] = 1.0f;
    }
    return mesh;
}

static lbvh_camera camera_at(int w, int h, float z, float yaw_deg = 0.0f)
{
    lbvh_camera cam;
    cam.screen_width = w; cam.screen_height = h;
    cam.camera_fov = (float)std::tan(60.0 * 3.14159265358979323846 / 180.0 / 2.0);   // Mathf.Tan(fov * Deg2Rad / 2)
    cam.near_plane = 0.3f;
    const float c = (float)std::cos(yaw_deg * 3.14159265358979323846 / 180.0), s = (float)std::sin(yaw_deg * 3.14159265358979323846 / 180.0);
    const float m[16] = {-c, 0, s, 0, 0, 1, 0, 0, s, 0, c, z, 0, 0, 0, 1};
    std::memcpy(cam.camera_to_world, m, sizeof m);
    return cam;
}

// BASELINE configs[2] through the one-process host: every frame of every camera / mode assembled from <ranks> shares must
// equal the frame one context traces alone, the assembled buffer poisoned before each frame
static int multi_main(int argc, char** argv)
{
    const int ranks = argc > 2 ? atoi(argv[2]) : 2;
    const uint32_t n = argc > 3 ? (uint32_t)atoi(argv[3]) : 4096;
    const int w = argc > 4 ? atoi(argv[4]) : 256, h = argc > 5 ? atoi(argv[5]) : 256;
    // "doubled": a scene full of exact t ties — LBVH_TRACE_FAST_EXACT's tie resolution then runs on EVERY rank against the owner's
    // frame (system-scope loads and compare-and-swaps on peer memory): ADVICE r4
    const bool doubled = argc > 6 && std::strcmp(argv[6], "doubled") == 0;
    const std::vector<lbvh_triangle> mesh = random_mesh(n, doubled);
    try {
        const int n_dev = lbvh::Context::device_count();
        if (n_dev <= 0) { std::fprintf(stderr, "no HIP device visible\n"); return 1; }
        std::vector<int> devices;
        for (int r = 0; r < ranks; r++) devices.push_back(r % n_dev);
        lbvh::MultiGpuDrawer multi(devices, mesh);
        multi.Awake();
        lbvh::Context one(0);
        lbvh::RaytracingMeshDrawer single(one, mesh);
        single.Awake();
        int frames = 0, equal = 0;
        size_t hits = 0;
        for (int mode : {LBVH_TRACE_FAST, LBVH_TRACE_REFERENCE, LBVH_TRACE_FAST_EXACT})
            for (int f = 0; f < 4; f++) {
                // two frames from one camera (the second is dispatched by the first one's costs), then a turned one, then a
                // rebuilt scene
                const lbvh_camera cam = camera_at(w, h, 300.0f, f < 2 ? 0.0f : 2.0f * f);
                if (f == 3) { multi.Rebuild(); single.Rebuild(); }
                if (frames) multi.Hits().Fill(0x7FC00000u, false);            // poison: a missing tile cannot pass on old values
                multi.Update(cam, mode);
                single.Update(cam, mode);
                multi.Hits().GetData();                                      // on the owner's stream: behind the device-side gather
                single.Hits().GetData();
                const bool same = std::memcmp(multi.Hits().LocalBuffer().data(), single.Hits().LocalBuffer().data(),
                                              (size_t)w * h * sizeof(lbvh_hit)) == 0;
                frames++;
                equal += same ? 1 : 0;
                if (mode == LBVH_TRACE_FAST_EXACT) {      // ... and the exact mode's frame is the reference mode's, word for word
                    // (include/lbvh.h: except at a pixel whose reference winner lies in front of its own triangle's box — this
                    // scene and these cameras have none; the general case is checked against the oracle by the Python suite)
                    single.Update(cam, LBVH_TRACE_REFERENCE);
                    single.Hits().GetData();
                    if (std::memcmp(multi.Hits().LocalBuffer().data(), single.Hits().LocalBuffer().data(), (size_t)w * h * sizeof(lbvh_hit)) != 0)
                        equal--;
                }
                if (mode == LBVH_TRACE_REFERENCE && f == 0)
                    for (size_t i = 0; i < (size_t)w * h; i++) hits += multi.Hits().LocalBuffer()[i].t < LBVH_MAX_FLOAT ? 1 : 0;
            }
        multi.Sync();
        std::printf("{\"ranks\": %d, \"devices\": %d, \"triangles\": %u, \"rays\": %d, \"frames\": %d, \"frames_equal\": %d, \"hits\": %zu}\n",
                    ranks, n_dev, n, w * h, frames, equal, hits);
        return equal == frames ? 0 : 2;
    } catch (const lbvh::Error& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
}

int main(int argc, char** argv)
{
    if (argc > 1 && std::strcmp(argv[1], "multi") == 0) return multi_main(argc, argv);
    const bool from_obj = argc > 2 && std::strcmp(argv[1], "obj") == 0;
    uint32_t n = !from_obj && argc > 1 ? (uint32_t)atoi(argv[1]) : 4096;
    const int w = from_obj ? (argc > 3 ? atoi(argv[3]) : 256) : (argc > 2 ? atoi(argv[2]) : 256);
    const int h = from_obj ? (argc > 4 ? atoi(argv[4]) : 256) : (argc > 3 ? atoi(argv[3]) : 256);
    const float cam_z = from_obj && argc > 5 ? (float)atof(argv[5]) : 300.0f;
    std::vector<lbvh_triangle> mesh;
    if (from_obj) {
        try {
            mesh = lbvh::MeshTriangles(lbvh::LoadObj(argv[2]));       // MeshBufferContainer.cs:117-146 for an OBJ asset
        } catch (const std::exception& e) {
            std::fprintf(stderr, "%s\n", e.what());
            return 1;
        }
        n = (uint32_t)mesh.size();
    } else {
        mesh.resize(n);
    }
    uint64_t seed = 1;
    for (auto& t : mesh) {
        if (from_obj) break;
        std::memset(&t, 0, sizeof t);
        for (int k = 0; k < 3; k++) {
            const float c = uniform(seed, -100.0f, 100.0f);
            t.a[k] = c;
            t.b[k] = c + uniform(seed, -2.0f, 2.0f);
            t.c[k] = c + uniform(seed, -2.0f, 2.0f);
        }
        t.b_uv[0] = 1.0f; t.c_uv[1] = 1.0f;
    }
    const bool dynamic = !from_obj && argc > 4 && std::strcmp(argv[4], "dynamic") == 0;
    try {
        lbvh::Context ctx(0);
        if (dynamic) {
            // BASELINE configs[4] in miniature: 8 rigid bodies (triangle i belongs to body i % 8), one animated frame,
            // primary rays + 2 bounces; prints a checksum of the RGBA16F image
            std::vector<uint32_t> body(n);
            for (uint32_t i = 0; i < n; i++) body[i] = i % 8;
            std::vector<float> centres(8 * 4, 0.0f);
            for (int bdy = 0; bdy < 8; bdy++) {
                centres[bdy * 4 + 0] = (bdy & 1) ? 40.0f : -40.0f;
                centres[bdy * 4 + 1] = (bdy & 2) ? 40.0f : -40.0f;
                centres[bdy * 4 + 2] = (bdy & 4) ? 40.0f : -40.0f;
            }
            lbvh::DynamicPathTracer pt(ctx, mesh, body, centres, 1e-3f, 0.7f, 3);
            pt.Animate(0.1f);
            lbvh_camera cam;
            cam.screen_width = w; cam.screen_height = h;
            cam.camera_fov = (float)std::tan(60.0 * 3.14159265358979323846 / 180.0 / 2.0);
            cam.near_plane = 0.3f;
            const float m[16] = {-1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 300, 0, 0, 0, 1};
            std::memcpy(cam.camera_to_world, m, sizeof m);
            pt.Render(cam, 2);
            pt.Image().GetData();
            uint64_t image_sum = 0;
            for (size_t i = 0; i < (size_t)w * h; i++) {
                const uint64_t px = pt.Image().LocalBuffer()[i];
                image_sum += (px & 0xFFFF) + ((px >> 16) & 0xFFFF) * 3 + ((px >> 32) & 0xFFFF) * 5 + ((px >> 48) & 0xFFFF) * 7;
            }
            std::printf("{\"triangles\": %u, \"rays\": %d, \"image_sum\": %llu}\n", n, w * h, (unsigned long long)image_sum);
            return 0;
        }
        lbvh::RaytracingMeshDrawer drawer(ctx, mesh);
        auto t0 = std::chrono::steady_clock::now();
        drawer.Awake();
        ctx.sync();
        auto t1 = std::chrono::steady_clock::now();
        lbvh_camera cam;
        cam.screen_width = w; cam.screen_height = h;
        cam.camera_fov = (float)std::tan(60.0 * 3.14159265358979323846 / 180.0 / 2.0);   // Mathf.Tan(fov * Deg2Rad / 2)
        cam.near_plane = 0.3f;
        const float m[16] = {-1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, cam_z, 0, 0, 0, 1};
        std::memcpy(cam.camera_to_world, m, sizeof m);
        lbvh::Event e0(ctx), e1(ctx);
        e0.record();
        drawer.Update(cam, LBVH_TRACE_FAST);
        e1.record();
        ctx.sync();
        auto t2 = std::chrono::steady_clock::now();
        const float update_device_ms = lbvh::Event::elapsed_ms(e0, e1);
        // the same frame as three shards traced one after the other into the one buffer: must reproduce it exactly
        drawer.Hits().GetData();
        const std::vector<lbvh_hit> whole = drawer.Hits().LocalBuffer();
        drawer.Hits().Fill(0xFFFFFFFFu);
        for (uint32_t r = 0; r < 3; r++) drawer.UpdateShard(cam, r, 3, LBVH_TRACE_FAST);
        drawer.Hits().GetData();
        const bool shards_equal = std::memcmp(whole.data(), drawer.Hits().LocalBuffer().data(), whole.size() * sizeof(lbvh_hit)) == 0;
        ctx.trace_forget();
        drawer.Update(cam, LBVH_TRACE_FAST);
        drawer.Container().GetAllGpuData();          // throws on a corrupted node (MeshBufferContainer.cs:181-195)
        drawer.Hits().GetData();
        uint64_t key_sum = 0, node_sum = 0;
        for (uint32_t i = 0; i < n; i++) key_sum += drawer.Container().Keys().LocalBuffer()[i];
        for (uint32_t i = 0; i + 1 < n; i++) {
            const auto& nd = drawer.Container().BvhInternalNode().LocalBuffer()[i];
            node_sum += (uint64_t)nd.leftNode * 3 + (uint64_t)nd.rightNode * 5 + (uint64_t)nd.parent * 7 + nd.leftNodeType + nd.rightNodeType;
        }
        size_t hits = 0;
        double tsum = 0;
        for (size_t i = 0; i < (size_t)w * h; i++)
            if (drawer.Hits().LocalBuffer()[i].t < LBVH_MAX_FLOAT) { hits++; tsum += drawer.Hits().LocalBuffer()[i].t; }
        std::printf("{\"triangles\": %u, \"rays\": %d, \"awake_ms\": %.3f, \"update_ms\": %.3f, \"update_device_ms\": %.4f, "
                    "\"key_sum\": %llu, \"node_sum\": %llu, \"hits\": %zu, \"t_sum\": %.6f, \"shards_equal\": %s}\n",
                    n, w * h, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                    std::chrono::duration<double, std::milli>(t2 - t1).count(), update_device_ms, (unsigned long long)key_sum,
                    (unsigned long long)node_sum, hits, tsum, shards_equal ? "true" : "false");
    } catch (const lbvh::Error& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
