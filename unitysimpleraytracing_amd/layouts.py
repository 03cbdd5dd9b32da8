"""Buffer element layouts of the hot path as numpy dtypes.

Bit-for-bit the reference's structs (Assets/_Shaders/Constants.cginc:9-54,
Assets/_Scripts/SceneDataTypes.cs:4-89) and include/lbvh.h.
"""
import numpy as np

AABB = np.dtype([("min", "<f4", 3), ("_dummy0", "<f4"), ("max", "<f4", 3), ("_dummy1", "<f4")])
TRIANGLE = np.dtype([
    ("a", "<f4", 3), ("_dummy0", "<f4"),
    ("b", "<f4", 3), ("_dummy1", "<f4"),
    ("c", "<f4", 3), ("_dummy2", "<f4"),
    ("a_uv", "<f4", 2), ("b_uv", "<f4", 2), ("c_uv", "<f4", 2), ("_dummy3", "<f4", 2),
    ("a_normal", "<f4", 3), ("_dummy4", "<f4"),
    ("b_normal", "<f4", 3), ("_dummy5", "<f4"),
    ("c_normal", "<f4", 3), ("_dummy6", "<f4"),
])
INTERNAL_NODE = np.dtype([
    ("leftNode", "<u4"), ("leftNodeType", "<u4"), ("rightNode", "<u4"), ("rightNodeType", "<u4"),
    ("parent", "<u4"), ("index", "<u4"),
])
LEAF_NODE = np.dtype([("parent", "<u4"), ("index", "<u4")])
HIT = np.dtype([("t", "<f4"), ("tri", "<u4"), ("u", "<f4"), ("v", "<f4")])
TRACE_STATS = np.dtype([("pops", "<u8"), ("box_hits", "<u8"), ("leaf_tests", "<u8"),
                        ("tri_tests", "<u8"), ("hits", "<u8")])

BUILD_FAST_SCENE, BUILD_RESET_NODES = 1, 2      # lbvh_build_scene flags

RAY_STATS = np.dtype([("rays", "<u8"), ("node_fetches", "<u8"), ("triangle_tests", "<u8")])
PATH_STATE = np.dtype([("origin", "<f4", 3), ("alive", "<u4"), ("dir", "<f4", 3), ("pad0", "<f4"),
                       ("throughput", "<f4", 3), ("pad1", "<f4"), ("radiance", "<f4", 3), ("alpha", "<f4")])
assert PATH_STATE.itemsize == 64

assert AABB.itemsize == 32          # Assets/_Scripts/MeshBufferContainer.cs:103
assert TRIANGLE.itemsize == 128     # Assets/_Scripts/MeshBufferContainer.cs:98
assert INTERNAL_NODE.itemsize == 24
assert LEAF_NODE.itemsize == 8
assert HIT.itemsize == 16

INTERNAL = 0                         # Assets/_Shaders/Constants.cginc:17
LEAF = 1                             # Assets/_Shaders/Constants.cginc:18
NULL = 0xFFFFFFFF                    # NullLeaf words, Assets/_Scripts/SceneDataTypes.cs:63-71
MAX_FLOAT = np.float32(2139095040.0)  # (float)0x7F7FFFFF, Assets/_Shaders/Constants.cginc:7

TRACE_REFERENCE = 0
TRACE_FAST = 1
TRACE_FAST_EXACT = 2      # TRACE_FAST + rays that meet an exact t tie re-traced by the reference's walk: == TRACE_REFERENCE, every word

# Scene box the reference hard-wires for Morton normalisation
# (Assets/_Scripts/MeshBufferContainer.cs:9-15)
SCENE_BOX_MIN = np.array([-125.0, -125.0, -125.0], dtype=np.float32)
SCENE_BOX_MAX = np.array([125.0, 125.0, 125.0], dtype=np.float32)
