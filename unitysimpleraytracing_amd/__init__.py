"""MI355X-native LBVH ray-tracing hot path (drop-in for drzhn/UnitySimpleRaytracing's
Morton -> radix sort -> DistributeKeys -> Karras tree -> refit -> primary-ray traversal).

The product is the C-ABI library ``liblbvh.so`` (``include/lbvh.h``), hand-written HIP for gfx950.
This package is the thin host side: ctypes bindings (`_native`), the reference's host classes
mirrored over the C ABI (`host`), buffer layouts (`layouts`) and synthetic scenes (`scenes`).
There is no CPU fallback: importing `_native` fails loudly if the library is missing.
"""

from . import layouts, scenes  # noqa: F401

__all__ = ["layouts", "scenes"]
