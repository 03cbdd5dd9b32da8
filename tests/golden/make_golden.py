"""Generates the committed golden fixtures under tests/golden/ with the CPU oracle.

The reference holds no golden vectors and cannot run here, so these are
oracle outputs on seeded inputs; the hand-derived KATs in test_oracle_kat.py and the literal sort
emulation are what pin the oracle itself.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle as O                                    # noqa: E402
from unitysimpleraytracing_amd import scenes          # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    # cfg1: 4 096 random triangles (seed 1), 64x64 primary rays from (0, 0, 300)
    tris = scenes.random_triangles(4096, seed=1)
    cap = scenes.capacity_for(len(tris))
    keys0, idx0, aabb0 = O.morton_aabb(tris, capacity=cap)
    b = O.Built(tris, capacity=cap)
    cam = scenes.camera(64, 64, (0.0, 0.0, 300.0))
    hits, stats = O.trace_primary(b, cam)
    np.savez_compressed(
        os.path.join(HERE, "cfg1_4096.npz"),
        positions=np.stack([tris["a"], tris["b"], tris["c"]], axis=1),   # (n, 3, 3) f32 — the input
        morton=keys0[:4096], sorted_keys=b.keys, sorted_indices=b.indices,
        internal=b.internal[:4095].view(np.uint32).reshape(-1, 6),
        leaf=b.leaf[:4096].view(np.uint32).reshape(-1, 2),
        bvh_min=b.bvh["min"][:4095], bvh_max=b.bvh["max"][:4095],
        hit_t=hits["t"], hit_tri=hits["tri"], hit_u=hits["u"], hit_v=hits["v"],
        stats=np.array([stats[f] for f in stats.dtype.names], dtype=np.uint64),
        camera_to_world=cam["camera_to_world"], camera_fov=np.float32(cam["camera_fov"]),
    )
    # reference default scene: 80x80 grid, camera of Scene.unity, 64x64 rays
    g = scenes.grid_scene()
    bg = O.Built(g, capacity=scenes.capacity_for(len(g)))
    camg = scenes.reference_scene_camera(64, 64)
    hg, sg = O.trace_primary(bg, camg)
    np.savez_compressed(
        os.path.join(HERE, "grid_80x80.npz"),
        sorted_keys=bg.keys[:12800], sorted_indices=bg.indices[:12800],
        internal=bg.internal[:12799].view(np.uint32).reshape(-1, 6),
        leaf=bg.leaf[:12800].view(np.uint32).reshape(-1, 2),
        bvh_min=bg.bvh["min"][:12799], bvh_max=bg.bvh["max"][:12799],
        hit_t=hg["t"], hit_tri=hg["tri"],
        stats=np.array([sg[f] for f in sg.dtype.names], dtype=np.uint64),
    )
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
