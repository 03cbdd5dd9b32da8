// lbvh_mesh.hpp — mesh ingest for the compiled host: what `new MeshBufferContainer(Mesh)` does with a
// Unity Mesh (Assets/_Scripts/MeshBufferContainer.cs:117-146), for Wavefront OBJ files such as the reference's own
// Assets/_Assets/*.obj.  Pure host code (no GPU, no HIP): lbvh::Mesh mirrors the four arrays the reference reads from
// UnityEngine.Mesh (vertices, triangles, uv, normals), lbvh::MeshTriangles is the reference's gather loop into 128-byte
// Triangle records, lbvh::LoadObj fills a Mesh from an OBJ.
//
// OBJ rules (identical to unitysimpleraytracing_amd/scenes.py load_obj, the Python twin; tests compare the two byte for
// byte): v / vt / vn records; faces with v, v/vt, v//vn or v/vt/vn corners; negative (relative) indices; a polygon
// a b c d ... is fanned a b c, a c d, ... in file order; numbers are parsed as doubles and rounded once to float.  When
// any corner of the file lacks a vt, every uv is (0,0); when any lacks a vn, every normal is the unit face normal
// cross(b - a, c - a) (fp32, each operation rounded; a zero-area face keeps the zero vector).  Unity's importer also
// mirrors x, flips the winding and may reorder triangles — an editor-side convention this loader does not imitate.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <istream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/lbvh.h"

namespace lbvh {

struct Float3 { float x, y, z; };
struct Float2 { float x, y; };

// The members of UnityEngine.Mesh that MeshBufferContainer's constructor reads (:117-121): per-vertex arrays and a flat
// index list, three entries per triangle.
struct Mesh {
    std::vector<Float3> vertices;
    std::vector<int> triangles;
    std::vector<Float2> uv;
    std::vector<Float3> normals;
};

// MeshBufferContainer.cs:123-146 without the Morton / AABB half (that is lbvh_morton_aabb on the device): one 128-byte
// Triangle per index triple, padding words zero.
inline std::vector<lbvh_triangle> MeshTriangles(const Mesh& mesh)
{
    const size_t n = mesh.triangles.size() / 3;
    std::vector<lbvh_triangle> out(n);
    if (n) std::memset(out.data(), 0, n * sizeof(lbvh_triangle));
    auto put3 = [](float* d, const Float3& s) { d[0] = s.x; d[1] = s.y; d[2] = s.z; };
    auto put2 = [](float* d, const Float2& s) { d[0] = s.x; d[1] = s.y; };
    for (size_t i = 0; i < n; i++) {
        const int ia = mesh.triangles[i * 3 + 0], ib = mesh.triangles[i * 3 + 1], ic = mesh.triangles[i * 3 + 2];
        lbvh_triangle& t = out[i];
        put3(t.a, mesh.vertices.at(ia)); put3(t.b, mesh.vertices.at(ib)); put3(t.c, mesh.vertices.at(ic));
        put2(t.a_uv, mesh.uv.at(ia)); put2(t.b_uv, mesh.uv.at(ib)); put2(t.c_uv, mesh.uv.at(ic));
        put3(t.a_normal, mesh.normals.at(ia)); put3(t.b_normal, mesh.normals.at(ib)); put3(t.c_normal, mesh.normals.at(ic));
    }
    return out;
}

namespace detail {
struct Corner { long v, t, n; };      // 0-based, -1 = absent

inline long resolve(const std::string& tok, size_t count, size_t line_no)
{
    if (tok.empty()) return -1;
    char* end = nullptr;
    const long k = std::strtol(tok.c_str(), &end, 10);
    if (*end != '\0' || k == 0) throw std::runtime_error("OBJ line " + std::to_string(line_no) + ": bad index '" + tok + "'");
    const long idx = k > 0 ? k - 1 : (long)count + k;
    if (idx < 0 || idx >= (long)count) throw std::runtime_error("OBJ line " + std::to_string(line_no) + ": index out of range");
    return idx;
}

inline float number(const std::string& tok, size_t line_no)
{
    char* end = nullptr;
    const double d = std::strtod(tok.c_str(), &end);
    if (end == tok.c_str() || *end != '\0') throw std::runtime_error("OBJ line " + std::to_string(line_no) + ": bad number '" + tok + "'");
    return (float)d;
}
}  // namespace detail

// One Mesh vertex per face corner (no welding: the reference's gather reads corners independently anyway).
inline Mesh ParseObj(std::istream& in)
{
    using detail::Corner;
    std::vector<Float3> v, vn;
    std::vector<Float2> vt;
    std::vector<Corner> corners;          // 3 per triangle, fanned
    std::string line;
    size_t line_no = 0;
    while (std::getline(in, line)) {
        line_no++;
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line.resize(hash);
        std::istringstream ls(line);
        std::vector<std::string> p;
        for (std::string tok; ls >> tok;) p.push_back(tok);
        if (p.empty()) continue;
        if (p[0] == "v" || p[0] == "vn") {
            if (p.size() < 4) throw std::runtime_error("OBJ line " + std::to_string(line_no) + ": three numbers expected");
            const Float3 f{detail::number(p[1], line_no), detail::number(p[2], line_no), detail::number(p[3], line_no)};
            (p[0] == "v" ? v : vn).push_back(f);
        } else if (p[0] == "vt") {
            if (p.size() < 2) throw std::runtime_error("OBJ line " + std::to_string(line_no) + ": a number expected");
            vt.push_back(Float2{detail::number(p[1], line_no), p.size() > 2 ? detail::number(p[2], line_no) : 0.0f});
        } else if (p[0] == "f") {
            std::vector<Corner> poly;
            for (size_t k = 1; k < p.size(); k++) {
                std::string part[3];
                size_t which = 0;
                for (char ch : p[k]) {
                    if (ch == '/') { if (++which > 2) break; }
                    else part[which] += ch;
                }
                poly.push_back(Corner{detail::resolve(part[0], v.size(), line_no), detail::resolve(part[1], vt.size(), line_no),
                                      detail::resolve(part[2], vn.size(), line_no)});
                if (poly.back().v < 0) throw std::runtime_error("OBJ line " + std::to_string(line_no) + ": corner without a vertex");
            }
            for (size_t k = 1; k + 1 < poly.size(); k++) {
                corners.push_back(poly[0]); corners.push_back(poly[k]); corners.push_back(poly[k + 1]);
            }
        }
    }
    bool all_uv = !vt.empty(), all_n = !vn.empty();
    for (const Corner& c : corners) { all_uv = all_uv && c.t >= 0; all_n = all_n && c.n >= 0; }
    Mesh m;
    m.vertices.reserve(corners.size()); m.uv.reserve(corners.size()); m.normals.reserve(corners.size());
    m.triangles.reserve(corners.size());
    for (size_t i = 0; i < corners.size(); i += 3) {
        Float3 face{0.0f, 0.0f, 0.0f};
        if (!all_n) {
            const Float3 &a = v[corners[i].v], &b = v[corners[i + 1].v], &c = v[corners[i + 2].v];
            const float e1x = b.x - a.x, e1y = b.y - a.y, e1z = b.z - a.z, e2x = c.x - a.x, e2y = c.y - a.y, e2z = c.z - a.z;
            // every product, difference and sum rounded to fp32 on its own (volatile keeps x87 / FMA builds honest)
            volatile float p0 = e1y * e2z, p1 = e1z * e2y, p2 = e1z * e2x, p3 = e1x * e2z, p4 = e1x * e2y, p5 = e1y * e2x;
            const float nx = p0 - p1, ny = p2 - p3, nz = p4 - p5;
            volatile float sx = nx * nx, sy = ny * ny, sz = nz * nz;
            volatile float s01 = sx + sy;
            float len = std::sqrt((float)(s01 + sz));
            if (len == 0.0f) len = 1.0f;
            face = Float3{nx / len, ny / len, nz / len};
        }
        for (size_t k = 0; k < 3; k++) {
            const Corner& c = corners[i + k];
            m.triangles.push_back((int)m.vertices.size());
            m.vertices.push_back(v[c.v]);
            m.uv.push_back(all_uv ? vt[c.t] : Float2{0.0f, 0.0f});
            m.normals.push_back(all_n ? vn[c.n] : face);
        }
    }
    return m;
}

inline Mesh LoadObj(const std::string& path)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open " + path);
    return ParseObj(in);
}

}  // namespace lbvh
