// Wavefront .obj loading for the host scene layer: TriMesh::load_as_vec, to_triangles and
// to_triangles_with_mat of src/objects/tri_mesh.rs:280-460.
//
// The reference delegates the parsing to tobj 4.0.2 (Cargo.toml:16, call at tri_mesh.rs:301
// with LoadOptions::default(): single_index = false, triangulate = false, points and lines
// kept).  That crate is not part of the reference tree, so what follows restates its
// published behaviour for those options:
//   - statements v / vt / vn / f / l / o / g / usemtl / mtllib, anything else ignored;
//     vt keeps two components; a `v` line may carry a colour, which is dropped here;
//   - face vertices are v, v/vt, v//vn or v/vt/vn, 1-based, negative = relative to the
//     elements read so far;
//   - a model ends at the next o / g line, or at a usemtl that names a different material
//     of the loaded .mtl, if it has faces by then; the last model ends with the file;
//   - per model the position / normal / texcoord arrays hold only what its faces use, in
//     order of first use, each with its own index stream (multi-index export);
//   - faces are not triangulated: a face of k vertices contributes k indices.  The reference
//     then reads the index stream three at a time (tri_mesh.rs:383-415), so only triangulated
//     files give meaningful triangles there, and the same holds here.
// Where the crate's handling of a malformed file is not known for certain (a face vertex
// without vt / vn in a model whose other vertices have them), loading fails, which the
// reference answers with "no meshes" (tri_mesh.rs:303-306).
#include <cctype>
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <unordered_map>

#include "pthost.hpp"

namespace pt {
namespace {

constexpr size_t MISSING = ~(size_t)0;

struct VertexIndices {
    size_t v, vt, vn;
};
using Face = std::vector<VertexIndices>;

std::vector<std::string> split_ws(const std::string &line) {
    std::vector<std::string> out;
    size_t i = 0;
    while (i < line.size()) {
        while (i < line.size() && std::isspace((unsigned char)line[i])) i++;
        size_t j = i;
        while (j < line.size() && !std::isspace((unsigned char)line[j])) j++;
        if (j > i) out.push_back(line.substr(i, j - i));
        i = j;
    }
    return out;
}
std::string trim(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) a++;
    while (b > a && std::isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}

// Decimal floats as Rust's f32::from_str takes them: [+-] digits [. digits] [e [+-] digits],
// at least one digit in the mantissa, or inf / infinity / nan (any case).  Correctly rounded.
bool parse_f32(const std::string &t, float &out) {
    size_t i = 0;
    if (i < t.size() && (t[i] == '+' || t[i] == '-')) i++;
    std::string rest = t.substr(i);
    for (auto &c : rest) c = (char)std::tolower((unsigned char)c);
    if (rest == "inf" || rest == "infinity" || rest == "nan") {
        out = std::strtof(t.c_str(), nullptr);
        return true;
    }
    size_t digits = 0;
    while (i < t.size() && std::isdigit((unsigned char)t[i])) i++, digits++;
    if (i < t.size() && t[i] == '.') {
        i++;
        while (i < t.size() && std::isdigit((unsigned char)t[i])) i++, digits++;
    }
    if (digits == 0) return false;
    if (i < t.size() && (t[i] == 'e' || t[i] == 'E')) {
        i++;
        if (i < t.size() && (t[i] == '+' || t[i] == '-')) i++;
        size_t ed = 0;
        while (i < t.size() && std::isdigit((unsigned char)t[i])) i++, ed++;
        if (ed == 0) return false;
    }
    if (i != t.size()) return false;
    out = std::strtof(t.c_str(), nullptr);
    return true;
}
// Appends n floats taken from words[first...]; false when fewer than n parse.
bool parse_floatn(const std::vector<std::string> &words, size_t first, size_t n, std::vector<float> &vals) {
    size_t got = 0;
    for (size_t k = first; k < words.size() && got < n; k++, got++) {
        float x;
        if (!parse_f32(words[k], x)) return false;
        vals.push_back(x);
    }
    return got == n;
}
bool parse_isize(const std::string &t, long long &out) {
    if (t.empty()) return false;
    size_t i = 0;
    if (t[i] == '+' || t[i] == '-') i++;
    if (i == t.size()) return false;
    for (size_t k = i; k < t.size(); k++)
        if (!std::isdigit((unsigned char)t[k])) return false;
    errno = 0;
    out = std::strtoll(t.c_str(), nullptr, 10);
    return errno == 0;
}
bool parse_vertex(const std::string &word, size_t n_pos, size_t n_tex, size_t n_nrm, VertexIndices &out) {
    size_t idx[3] = {MISSING, MISSING, MISSING};
    size_t field = 0, start = 0;
    for (;;) {
        const size_t slash = word.find('/', start);
        const std::string part = word.substr(start, slash == std::string::npos ? std::string::npos : slash - start);
        if (!part.empty()) {
            if (field > 2) return false;
            long long x;
            if (!parse_isize(part, x)) return false;
            const size_t count = field == 0 ? n_pos : (field == 1 ? n_tex : n_nrm);
            idx[field] = x < 0 ? (size_t)((long long)count + x) : (size_t)(x - 1);
        }
        if (slash == std::string::npos) break;
        start = slash + 1;
        field++;
    }
    out = VertexIndices{idx[0], idx[1], idx[2]};
    return true;
}

struct Pools {
    std::vector<float> pos, tex, nrm;
};

// Multi-index export of the faces gathered for one model.
TriMesh export_faces(const Pools &p, const std::vector<Face> &faces) {
    TriMesh m;
    std::unordered_map<size_t, uint32_t> vmap, tmap, nmap;
    size_t with_vt = 0, with_vn = 0, total = 0;
    for (const Face &f : faces)
        for (const VertexIndices &vi : f) {
            total++;
            with_vt += vi.vt != MISSING;
            with_vn += vi.vn != MISSING;
        }
    const bool use_vt = !p.tex.empty() && with_vt != 0, use_vn = !p.nrm.empty() && with_vn != 0;
    if ((use_vt && with_vt != total) || (use_vn && with_vn != total))
        throw std::runtime_error("some face vertices of a model have a texcoord/normal index and others do not");
    for (const Face &f : faces)
        for (const VertexIndices &vi : f) {
            auto it = vmap.find(vi.v);
            if (it != vmap.end()) {
                m.position_indices.push_back(it->second);
            } else {
                if (vi.v >= p.pos.size() / 3) throw std::runtime_error("face vertex index out of bounds");
                const uint32_t next = (uint32_t)vmap.size();
                m.positions.push_back(Point3(p.pos[3 * vi.v], p.pos[3 * vi.v + 1], p.pos[3 * vi.v + 2]));
                m.position_indices.push_back(next);
                vmap.emplace(vi.v, next);
            }
            if (use_vt) {
                auto jt = tmap.find(vi.vt);
                if (jt != tmap.end()) {
                    m.uv_indices.push_back(jt->second);
                } else {
                    if (vi.vt >= p.tex.size() / 2) throw std::runtime_error("face texcoord index out of bounds");
                    const uint32_t next = (uint32_t)tmap.size();
                    m.uvs.push_back({p.tex[2 * vi.vt], p.tex[2 * vi.vt + 1]});
                    m.uv_indices.push_back(next);
                    tmap.emplace(vi.vt, next);
                }
            }
            if (use_vn) {
                auto kt = nmap.find(vi.vn);
                if (kt != nmap.end()) {
                    m.normal_indices.push_back(kt->second);
                } else {
                    if (vi.vn >= p.nrm.size() / 3) throw std::runtime_error("face normal index out of bounds");
                    const uint32_t next = (uint32_t)nmap.size();
                    m.normals.push_back(Vector3(p.nrm[3 * vi.vn], p.nrm[3 * vi.vn + 1], p.nrm[3 * vi.vn + 2]));
                    m.normal_indices.push_back(next);
                    nmap.emplace(vi.vn, next);
                }
            }
        }
    m.has_normals = !m.normals.empty();  // `mesh_obj.normals.len() > 0`, tri_mesh.rs:351
    m.has_uvs = !m.uvs.empty();          // tri_mesh.rs:363
    return m;
}

// newmtl names of a .mtl file, in order; false when the file cannot be read.
bool load_mtl_names(const std::string &path, std::unordered_map<std::string, int> &out) {
    std::ifstream f(path);
    if (!f) return false;
    std::string line;
    while (std::getline(f, line)) {
        const auto w = split_ws(line);
        if (w.size() >= 1 && w[0] == "newmtl") {
            const std::string name = trim(line.substr(line.find("newmtl") + 6));
            if (!name.empty() && !out.count(name)) {
                const int id = (int)out.size();
                out.emplace(name, id);
            }
        }
    }
    return true;
}

}  // namespace

std::vector<TriMesh> TriMesh::load_as_vec(const std::string &path, std::string *error) {
    std::vector<TriMesh> models;
    try {
        std::ifstream file(path);
        if (!file) throw std::runtime_error("cannot open " + path);
        const size_t dir_end = path.find_last_of('/');
        const std::string dir = dir_end == std::string::npos ? std::string() : path.substr(0, dir_end + 1);
        Pools pools;
        std::vector<Face> faces;
        std::unordered_map<std::string, int> mat_map;
        int mat_id = -1;  // None
        std::string line;
        while (std::getline(file, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            const auto words = split_ws(line);
            if (words.empty() || words[0] == "#") continue;
            const std::string &kw = words[0];
            if (kw == "v") {
                if (!parse_floatn(words, 1, 3, pools.pos)) throw std::runtime_error("position parse error");
            } else if (kw == "vt") {
                if (!parse_floatn(words, 1, 2, pools.tex)) throw std::runtime_error("texcoord parse error");
            } else if (kw == "vn") {
                if (!parse_floatn(words, 1, 3, pools.nrm)) throw std::runtime_error("normal parse error");
            } else if (kw == "f" || kw == "l") {
                Face face;
                for (size_t k = 1; k < words.size(); k++) {
                    VertexIndices vi;
                    if (!parse_vertex(words[k], pools.pos.size() / 3, pools.tex.size() / 2, pools.nrm.size() / 3, vi))
                        throw std::runtime_error("face parse error");
                    face.push_back(vi);
                }
                if (face.empty()) throw std::runtime_error("face parse error");
                faces.push_back(std::move(face));
            } else if (kw == "o" || kw == "g") {
                if (!faces.empty()) {
                    models.push_back(export_faces(pools, faces));
                    faces.clear();
                }
            } else if (kw == "mtllib") {
                const std::string name = trim(line.substr(line.find("mtllib") + 6));
                std::unordered_map<std::string, int> loaded;
                if (load_mtl_names(dir + name, loaded))  // an unreadable .mtl leaves the material table empty (tri_mesh.rs:309-318)
                    for (auto &kv : loaded)
                        if (!mat_map.count(kv.first)) {
                            const int id = (int)mat_map.size();
                            mat_map.emplace(kv.first, id);
                        }
            } else if (kw == "usemtl") {
                const std::string name = trim(line.substr(line.find("usemtl") + 6));
                if (name.empty()) throw std::runtime_error("material parse error");
                auto it = mat_map.find(name);
                const int new_mat = it == mat_map.end() ? -1 : it->second;
                if (new_mat != mat_id && !faces.empty()) {
                    models.push_back(export_faces(pools, faces));
                    faces.clear();
                }
                mat_id = new_mat;
            }
        }
        models.push_back(export_faces(pools, faces));  // the last model ends with the file
    } catch (const std::exception &e) {
        if (error) *error = e.what();
        return {};  // "Using empty mesh instead", tri_mesh.rs:303-306
    }
    if (error) error->clear();
    return models;
}

std::vector<Triangle> TriMesh::to_triangles_with_mat(MaterialRef material) const {
    std::vector<Triangle> out;
    const size_t range = position_indices.size();
    // (0..range).step_by(3) with [i], [i + 1], [i + 2] (tri_mesh.rs:386-394): a tail of one or two
    // indices would index out of range there; it is refused here
    if (range % 3 != 0) throw std::runtime_error("index stream length is not a multiple of three (the file is not triangulated)");
    out.reserve(range / 3);
    for (size_t i = 0; i < range; i += 3) {
        const std::array<Point3, 3> v = {positions[position_indices[i]], positions[position_indices[i + 1]], positions[position_indices[i + 2]]};
        std::array<Vector3, 3> n;
        std::array<std::array<float, 2>, 3> t;
        if (has_normals) n = {normals[normal_indices[i]], normals[normal_indices[i + 1]], normals[normal_indices[i + 2]]};
        if (has_uvs) t = {uvs[uv_indices[i]], uvs[uv_indices[i + 1]], uvs[uv_indices[i + 2]]};
        out.emplace_back(v, has_normals ? &n : nullptr, has_uvs ? &t : nullptr, material);
    }
    return out;
}
std::vector<Triangle> TriMesh::to_triangles() const { return to_triangles_with_mat(MaterialRef()); }

}  // namespace pt

// ------------------------------------------------------------------ C entry points
#include "../../include/pthost.h"

struct pth_obj {
    std::vector<pt::TriMesh> meshes;
};

extern "C" {

pth_obj *pth_obj_load(const char *path, char *err, size_t errlen) {
    if (err && errlen) err[0] = 0;
    if (!path) return nullptr;
    std::string why;
    auto meshes = pt::TriMesh::load_as_vec(path, &why);
    if (!why.empty()) {
        if (err && errlen) std::snprintf(err, errlen, "%s", why.c_str());
        return nullptr;
    }
    pth_obj *o = new (std::nothrow) pth_obj();
    if (o) o->meshes = std::move(meshes);
    return o;
}
void pth_obj_free(pth_obj *o) { delete o; }
uint32_t pth_obj_mesh_count(const pth_obj *o) { return o ? (uint32_t)o->meshes.size() : 0u; }

int pth_obj_mesh_get(const pth_obj *o, uint32_t i, pth_obj_mesh *out) {
    if (!o || !out || i >= o->meshes.size()) return 1;
    const pt::TriMesh &m = o->meshes[i];
    static_assert(sizeof(pt::Point3) == 12, "positions are exposed as packed floats");
    out->positions = m.positions.empty() ? nullptr : &m.positions[0].x;
    out->n_positions = (uint32_t)m.positions.size();
    out->indices = m.position_indices.data();
    out->n_indices = (uint32_t)m.position_indices.size();
    out->normals = m.normals.empty() ? nullptr : &m.normals[0].x;
    out->n_normals = (uint32_t)m.normals.size();
    out->normal_indices = m.normal_indices.data();
    out->texcoords = m.uvs.empty() ? nullptr : m.uvs[0].data();
    out->n_texcoords = (uint32_t)m.uvs.size();
    out->texcoord_indices = m.uv_indices.data();
    return 0;
}

int64_t pth_obj_triangles(const pth_obj *o, uint32_t i, float *pos, float *nrm, float *uv, int *has_normals, int *has_uvs) {
    if (!o || i >= o->meshes.size()) return -1;
    const pt::TriMesh &m = o->meshes[i];
    if (m.position_indices.size() % 3 != 0) return -1;
    const size_t n = m.position_indices.size() / 3;
    if (has_normals) *has_normals = m.has_normals;
    if (has_uvs) *has_uvs = m.has_uvs;
    for (size_t t = 0; t < n; t++)
        for (int c = 0; c < 3; c++) {
            const size_t k = 3 * t + c;
            if (pos) {
                const pt::Point3 &p = m.positions[m.position_indices[k]];
                pos[3 * k] = p.x; pos[3 * k + 1] = p.y; pos[3 * k + 2] = p.z;
            }
            if (nrm && m.has_normals) {
                const pt::Vector3 &v = m.normals[m.normal_indices[k]];
                nrm[3 * k] = v.x; nrm[3 * k + 1] = v.y; nrm[3 * k + 2] = v.z;
            }
            if (uv && m.has_uvs) {
                uv[2 * k] = m.uvs[m.uv_indices[k]][0];
                uv[2 * k + 1] = m.uvs[m.uv_indices[k]][1];
            }
        }
    return (int64_t)n;
}

}  // extern "C"
