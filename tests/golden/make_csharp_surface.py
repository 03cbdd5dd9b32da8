#!/usr/bin/env python3
"""Writes tests/golden/reference_csharp_surface.json: the public surface (constructors, methods, properties, serialized
fields, Unity messages — names and types only, no code) of the five reference classes SURVEY 8(b) names as the drop-in
boundary, scanned from /root/reference/Assets/_Scripts.  Run in the build container (the reference tree is not on the GPU
box); tests/test_csharp_surface.py compares bindings/csharp/*.Native.cs against this fixture."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from csharp_surface import surface          # noqa: E402

REF = "/root/reference/Assets/_Scripts"
FILES = ["DataBuffer.cs", "MeshBufferContainer.cs", "ComputeBufferSorter.cs", "BVHConstructor.cs", "RaytracingMeshDrawer.cs"]

if __name__ == "__main__":
    out = {}
    for f in FILES:
        out[f] = surface(os.path.join(REF, f))
    with open(os.path.join(HERE, "reference_csharp_surface.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print(json.dumps(out, indent=1, sort_keys=True))
