"""Generates the committed golden fixtures under tests/golden/ with the INDEPENDENT literal emulation
(oracle/literal_emulation.py: Python, one emulated GPU thread per dispatch id, written from the HLSL / C# text) —
NOT with oracle/lbvh_oracle.c.  The C oracle and the GPU library are then both checked against these files
(tests/test_oracle_kat.py, tests/test_gpu_parity.py): two independently written restatements + the HIP kernels
agreeing bit for bit is the strongest pin this reference allows (it holds no golden vectors and cannot run here;
parity stays "unpinned" by the letter — DESIGN.md section 2).

Inputs:
  cfg1_4096            BASELINE configs[0]: 4 096 random triangles (seed 1), 64x64 primary rays from (0, 0, 300)
  grid_80x80           the reference's default mesh re-created procedurally (80x80 quad grid), camera of Scene.unity
  example_object3      the reference's own mesh asset Assets/_Assets/ExampleObject3.obj (wired at Scene.unity:364),
                       parsed HERE from /root/reference (the asset is data; the file itself is not copied): OBJ face
                       order, quads fanned (a b c, a c d).  Unity's importer additionally mirrors x, reverses the
                       winding and may reorder triangles (meshOptimizationFlags -1): not reproducible offline, and
                       irrelevant to what the path computes per triangle.
  viking_room          Assets/_Assets/viking_room.obj + viking_room.png (the textured asset of the reference):
                       triangles with uv / normals, the texture box-filtered 4x4 to 256x256 RGBA8 (integer arithmetic),
                       64x64 primary rays, and the shaded RGBA16F image (Raytracing.compute:178-184).  The mesh spans
                       about +-0.7 units inside the +-125 Morton box, so nearly all Morton codes coincide: the fixture
                       that exercises DistributeKeys' duplicate handling and deep, degenerate trees.
Run from the repo root in the build container (needs /root/reference for the two asset fixtures):
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import literal_emulation as E             # noqa: E402
from unitysimpleraytracing_amd import scenes          # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ASSETS = "/root/reference/Assets/_Assets"


def build_and_trace(tris, cam, cap):
    r = E.awake(tris["a"], tris["b"], tris["c"], capacity=cap)
    t, tri, u, v, st = E.raytracing(r["scene"], cam)
    n = r["n"]
    out = dict(
        morton=r["morton"], sorted_keys=r["sorted_keys"], sorted_indices=r["sorted_indices"],
        internal=r["internal"][: n - 1], leaf=r["leaf"][:n], bvh_min=r["bvh_min"], bvh_max=r["bvh_max"],
        tri_min=r["tri_min"], tri_max=r["tri_max"],
        hit_t=t, hit_tri=tri, hit_u=u, hit_v=v, stats=np.array(st, dtype=np.uint64),
        camera_to_world=np.asarray(cam["camera_to_world"], dtype=np.float32), camera_fov=np.float32(cam["camera_fov"]),
        camera_near=np.float32(cam["near_plane"]), resolution=np.array([cam["screen_width"], cam["screen_height"]]),
    )
    return r, out


def downsample4(rgb):
    """(1024, 1024, 3|4) uint8 -> (256, 256, 4) uint8: 4x4 box filter, round half up, alpha 255"""
    h, w = rgb.shape[0] // 4, rgb.shape[1] // 4
    acc = rgb[: 4 * h, : 4 * w, :3].astype(np.uint32).reshape(h, 4, w, 4, 3).sum(axis=(1, 3))
    out = np.full((h, w, 4), 255, dtype=np.uint8)
    out[..., :3] = ((acc + 8) // 16).astype(np.uint8)
    return out


def main():
    tris = scenes.random_triangles(4096, seed=1)
    cam = scenes.camera(64, 64, (0.0, 0.0, 300.0))
    _, out = build_and_trace(tris, cam, scenes.capacity_for(len(tris)))
    np.savez_compressed(os.path.join(HERE, "cfg1_4096.npz"),
                        positions=np.stack([tris["a"], tris["b"], tris["c"]], axis=1), **out)

    g = scenes.grid_scene()
    _, out = build_and_trace(g, scenes.reference_scene_camera(64, 64), scenes.capacity_for(len(g)))
    np.savez_compressed(os.path.join(HERE, "grid_80x80.npz"), **out)

    if not os.path.isdir(REF_ASSETS):
        print("no /root/reference here: the asset fixtures are left as committed")
        return
    ex = scenes.load_obj(os.path.join(REF_ASSETS, "ExampleObject3.obj"))
    _, out = build_and_trace(ex, scenes.reference_scene_camera(64, 64), scenes.capacity_for(len(ex)))
    np.savez_compressed(os.path.join(HERE, "example_object3.npz"), triangles=ex, **out)

    from PIL import Image
    vk = scenes.load_obj(os.path.join(REF_ASSETS, "viking_room.obj"))
    png = np.asarray(Image.open(os.path.join(REF_ASSETS, "viking_room.png")).convert("RGB"))
    tex = downsample4(png[::-1])                       # row 0 at v = 0 (Unity's convention; PNG rows run top-down)
    cam = scenes.camera(64, 64, (0.05, 0.0, 1.6))
    r, out = build_and_trace(vk, cam, scenes.capacity_for(len(vk)))
    img = np.zeros((64, 64, 4), dtype=np.float16)
    for j in range(64):
        for i in range(64):
            res = [out["hit_t"][j, i], int(out["hit_tri"][j, i]), out["hit_u"][j, i], out["hit_v"][j, i]]
            img[j, i] = E.shade(res, vk[res[1]], tex)
    np.savez_compressed(os.path.join(HERE, "viking_room.npz"), triangles=vk, texture=tex, shaded=img.view(np.uint16), **out)
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
