"""SURVEY 8(f) rank 4 / VERDICT r2 item 6: the re-hosted C# classes under bindings/csharp/ keep the reference's public
surface.  No C# toolchain exists in the image, so the check is structural: both sides are scanned with
tests/csharp_surface.py and compared member by member — ComputeBuffer reads NativeBuffer on this side, nothing else may
differ.  The reference's side is a committed fixture (tests/golden/reference_csharp_surface.json, written by
tests/golden/make_csharp_surface.py from /root/reference); when the reference tree is present the fixture itself is
checked against a fresh scan."""
import json
import os
import re

import pytest

from csharp_surface import surface

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "bindings", "csharp")
FIXTURE = os.path.join(ROOT, "tests", "golden", "reference_csharp_surface.json")
REF = "/root/reference/Assets/_Scripts"

# what the native host adds on purpose (each justified in its file's header) / leaves out on purpose
ADDED = {
    "MeshBufferContainer": {"LbvhNative.Scene NativeScene()"},                    # the six buffers Update() binds, as one struct
    "ComputeBufferSorter<TKey,TValue>": {"bool ValidateAfterSort"},               # the reference validates always; here opt-in
}
DROPPED = {
    "RaytracingMeshDrawer": {"message OnDrawGizmos()"},                           # editor gizmos: SURVEY section 2, out of scope
}


def native(member):
    return re.sub(r"\bComputeBuffer\b", "NativeBuffer", member)


def reference_surface():
    return json.load(open(FIXTURE))


@pytest.mark.parametrize("ref_file", ["DataBuffer.cs", "MeshBufferContainer.cs", "ComputeBufferSorter.cs", "BVHConstructor.cs",
                                      "RaytracingMeshDrawer.cs"])
def test_rehosted_class_keeps_the_reference_surface(ref_file):
    ref = reference_surface()[ref_file]
    ours = surface(os.path.join(CS, ref_file.replace(".cs", ".Native.cs")))
    assert set(ours) == set(ref), (sorted(ours), sorted(ref))
    for cls, members in ref.items():
        want = {native(m) for m in members} - DROPPED.get(cls, set())
        got = set(ours[cls]) - ADDED.get(cls, set())
        assert got == want, {"missing": sorted(want - got), "extra": sorted(got - want)}
        assert ADDED.get(cls, set()) <= set(ours[cls])


def test_fixture_matches_the_reference_tree_when_it_is_present():
    if not os.path.isdir(REF):
        pytest.skip("the reference tree is not on this machine (GPU box): the committed fixture stands")
    for f, classes in reference_surface().items():
        assert surface(os.path.join(REF, f)) == classes, f


def test_scanner_on_a_known_snippet():
    src = '''
    public class Foo<T> : IDisposable where T : struct {
        [SerializeField] private Mesh _mesh;
        public ComputeBuffer Buf => _b;            // property
        private readonly int _x;
        public Foo(int size, T v) : this(size) { for (int i = 0; i < size; i++) { } }
        public Foo(int size) { }
        public T this[uint i] { get { return a[i]; } set { a[i] = value; } }
        public void Sort() { if (x) { y(); } }
        void Awake() { }
        private void Helper(int a) { }
        public static uint Count(uint[] keys, int n = 3) => 0;
    }'''
    assert surface(src, is_text=True) == {"Foo<T>": sorted([
        "serialized Mesh _mesh", "ComputeBuffer Buf", "ctor(int, T)", "ctor(int)", "T this[uint]", "void Sort()", "message Awake()",
        "uint Count(uint[], int)"])}


def test_native_binding_needs_no_unsafe_code():
    """LbvhNative.Camera used `unsafe fixed`, which Unity compiles only with "allow unsafe code"."""
    for f in os.listdir(CS):
        text = re.sub(r"//[^\n]*", "", open(os.path.join(CS, f)).read())
        assert not re.search(r"\bunsafe\b|\bfixed\b", text), f


def test_camera_struct_matches_the_c_layout():
    """LbvhNative.Camera = lbvh_camera: 2 ints, 2 floats, 16 matrix floats in row-major order, 80 bytes."""
    text = open(os.path.join(CS, "LbvhNative.cs")).read()
    body = re.search(r"public struct Camera\s*\{(.*?)\}", text, flags=re.S).group(1)
    fields = []
    for typ, names in re.findall(r"public\s+(int|float)\s+([^;]+);", body):
        fields += [(typ, n.strip()) for n in names.split(",")]
    assert [t for t, _ in fields] == ["int"] * 2 + ["float"] * 18
    assert [n for _, n in fields[4:]] == [f"m{r}{c}" for r in range(4) for c in range(4)]
    from unitysimpleraytracing_amd import _native as N
    import ctypes as C
    assert C.sizeof(N.Camera) == 4 * len(fields) == 80
