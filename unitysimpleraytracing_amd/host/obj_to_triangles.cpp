// obj_to_triangles <in.obj> <out.bin>: the compiled host's OBJ ingest on its own — lbvh::LoadObj + lbvh::MeshTriangles
// (lbvh_mesh.hpp), the 128-byte Triangle records written raw.  Host only (no GPU); tests/test_scenes.py compares the
// file with the Python twin's (scenes.load_obj) byte for byte.
#include <cstdio>

#include "lbvh_mesh.hpp"

int main(int argc, char** argv)
{
    if (argc != 3) { std::fprintf(stderr, "usage: %s in.obj out.bin\n", argv[0]); return 2; }
    try {
        const std::vector<lbvh_triangle> tris = lbvh::MeshTriangles(lbvh::LoadObj(argv[1]));
        FILE* f = std::fopen(argv[2], "wb");
        if (!f) { std::fprintf(stderr, "cannot write %s\n", argv[2]); return 1; }
        if (!tris.empty() && std::fwrite(tris.data(), sizeof(lbvh_triangle), tris.size(), f) != tris.size()) { std::fclose(f); return 1; }
        std::fclose(f);
        std::printf("%zu triangles\n", tris.size());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
