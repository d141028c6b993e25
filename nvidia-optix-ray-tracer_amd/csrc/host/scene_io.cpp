// scene_io.cpp -- host-side readers of the reference's input formats (include/hrt_io.h), plain C++17, no
// VTK / nlohmann dependency.  Each function cites the reference code whose observable result it reproduces.
#include "hrt_io.h"
#include "json_min.hpp"
#include "../cr_trig.h"      // cos / sin of constructRotateMatrix: the correctly rounded pin the pose kernel and the oracle share

#include <algorithm>
#include <array>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <map>
#include <string>
#include <vector>

namespace {

thread_local std::string g_error;

int fail(const std::string &msg) { g_error = msg; return -1; }

bool read_text(const std::string &path, std::string &out) {
    std::ifstream f(path, std::ios::in | std::ios::binary);
    if (!f) return false;
    std::ostringstream ss;
    ss << f.rdbuf();
    out = ss.str();
    return true;
}

template <typename T> T *dup_array(const std::vector<T> &v) {
    T *p = static_cast<T *>(std::malloc(std::max<size_t>(1, v.size() * sizeof(T))));
    if (p && !v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}
char *dup_string(const std::string &s) {
    char *p = static_cast<char *>(std::malloc(s.size() + 1));
    if (p) std::memcpy(p, s.c_str(), s.size() + 1);
    return p;
}

// whitespace-separated tokens of a text file, with the byte offset of each
struct Tokens {
    const std::string &s; size_t p = 0;
    explicit Tokens(const std::string &text) : s(text) {}
    bool next(std::string &tok) {
        while (p < s.size() && std::isspace((unsigned char)s[p])) ++p;
        if (p >= s.size()) return false;
        const size_t b = p;
        while (p < s.size() && !std::isspace((unsigned char)s[p])) ++p;
        tok.assign(s, b, p - b);
        return true;
    }
    bool number(double &d) {
        std::string t;
        if (!next(t)) return false;
        char *e = nullptr;
        d = std::strtod(t.c_str(), &e);
        return e != t.c_str() && *e == '\0';
    }
    void skip_line() { while (p < s.size() && s[p] != '\n') ++p; }
};


// Blocks a legacy VTK writer may add and the reference's vtkPolyDataReader reads past: FIELD <name> <n> followed by n arrays
// "<name> <components> <tuples> <type>" + values, and METADATA ... up to the next empty line.  Returns false on a malformed block.
bool skip_vtk_block(Tokens &tk, const std::string &keyword) {
    if (keyword == "METADATA") {
        while (tk.p < tk.s.size()) {
            tk.skip_line();
            if (tk.p < tk.s.size()) ++tk.p;
            size_t q = tk.p;
            while (q < tk.s.size() && (tk.s[q] == ' ' || tk.s[q] == '\t' || tk.s[q] == '\r')) ++q;
            if (q >= tk.s.size() || tk.s[q] == '\n') break;          // the blank line that ends the block
        }
        return true;
    }
    if (keyword == "FIELD") {
        std::string name; double n_arrays;
        if (!tk.next(name) || !tk.number(n_arrays) || n_arrays < 0 || n_arrays > 1e6) return false;
        for (uint64_t a = 0; a < (uint64_t)n_arrays; ++a) {
            std::string an, type; double comps, tuples;
            if (!tk.next(an) || !tk.number(comps) || !tk.number(tuples) || !tk.next(type) || comps < 0 || tuples < 0 || comps * tuples > 1e12) return false;
            const uint64_t count = (uint64_t)comps * (uint64_t)tuples;
            std::string v;
            for (uint64_t i = 0; i < count; ++i) if (!tk.next(v)) return false;
            // an array may carry its own METADATA block
            const size_t save = tk.p;
            if (tk.next(v) && v == "METADATA") skip_vtk_block(tk, v); else tk.p = save;
        }
        return true;
    }
    return false;
}

std::string lower(std::string v) { for (auto &c : v) c = (char)std::tolower((unsigned char)c); return v; }

// ---- constructTransformMatrix, include/Global/DeviceFunctions.cuh:43-148 (host side, float libm) ----
struct Mat4 { float m[4][4]; };
Mat4 mul(const Mat4 &a, const Mat4 &b) {
    Mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float sum = 0.0f;
            for (int n = 0; n < 4; ++n) sum += a.m[i][n] * b.m[n][j];
            r.m[i][j] = sum;
        }
    return r;
}
Mat4 ident() { Mat4 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = i == j ? 1.0f : 0.0f; return r; }
Mat4 rotation(float degree, int axis) {
    const float pi = 3.1415926f;
    const float theta = degree * pi / 180.0f;
    const float c = hrt::cosf_cr(theta), s = hrt::sinf_cr(theta);
    Mat4 r = ident();
    if (axis == 0) { r.m[1][1] = c; r.m[1][2] = -s; r.m[2][1] = s; r.m[2][2] = c; }
    else if (axis == 1) { r.m[0][0] = c; r.m[0][2] = s; r.m[2][0] = -s; r.m[2][2] = c; }
    else { r.m[0][0] = c; r.m[0][1] = -s; r.m[1][0] = s; r.m[1][1] = c; }
    return r;
}
void construct_transform(const float *shift, const float *rot, const float *scale, float *out) {
    Mat4 s = ident(), sc = ident();
    s.m[0][3] = shift[0]; s.m[1][3] = shift[1]; s.m[2][3] = shift[2];
    sc.m[0][0] = scale[0]; sc.m[1][1] = scale[1]; sc.m[2][2] = scale[2];
    const Mat4 r = mul(mul(rotation(rot[0], 0), rotation(rot[1], 1)), rotation(rot[2], 2));
    const Mat4 t = mul(mul(s, r), sc);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) out[4 * i + j] = t.m[i][j];
}


// ---- vtkPolyDataNormals' orientation pass (Consistency + AutoOrientNormals, as the reference sets them:
// src/Util/VTKReaderImpl.cpp:54-60 for Mesh files, :279-285 for STL shapes).  VTK's published algorithm, restated: per
// connected component the seed is found at the leftmost point (smallest x) that still has unvisited triangles -- of those, the
// one whose normal is most aligned with the x axis -- and is reversed when its normal points right (+x); orientation then
// spreads through MANIFOLD edges (exactly one neighbour): a neighbour that walks the shared edge in the same direction is
// reversed.  Returns one flag per triangle: reversed.  The reference takes the VERTICES from the reader's output and only the
// NORMALS from this filter, so a reversed triangle keeps its file winding and gets the opposite normal.
// Not reproducible from the published description: the order in which VTK's priority queue pops points of EQUAL x (here: the
// lowest point id first).  tri_points: 3 point ids per triangle; pts: 3 doubles per point.
std::vector<uint8_t> orient_triangles(const std::vector<uint64_t> &tri_points, const std::vector<double> &pts) {
    const size_t nt = tri_points.size() / 3, np = pts.size() / 3;
    std::vector<uint8_t> flip(nt, 0), visited(nt, 0);
    if (nt == 0) return flip;
    std::vector<std::vector<uint32_t>> cells_of(np);
    for (size_t t = 0; t < nt; ++t)
        for (int k = 0; k < 3; ++k) {
            auto &v = cells_of[tri_points[3 * t + k]];
            if (v.empty() || v.back() != (uint32_t)t) v.push_back((uint32_t)t);
        }
    std::vector<uint64_t> cur(tri_points);
    auto reverse_cell = [&](size_t t) { std::swap(cur[3 * t], cur[3 * t + 2]); flip[t] ^= 1u; };
    auto unit_normal_x = [&](size_t t) -> double {
        const double *A = &pts[3 * cur[3 * t]], *B = &pts[3 * cur[3 * t + 1]], *C = &pts[3 * cur[3 * t + 2]];
        const double e1[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, e2[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
        const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        const double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        return len > 0.0 ? n[0] / len : 0.0;
    };
    // the triangles other than t that use both p and q: from a table of the edges (unordered point pairs; reversing a triangle does not
    // change them) sorted once -- scanning the triangles around p instead is quadratic in the valence, and a fan of 10^5 triangles
    // around one merged point (untrusted files are parsed here) took minutes
    struct Edge { uint64_t a, b; uint32_t t; };
    std::vector<Edge> edges;
    edges.reserve(3 * nt);
    for (size_t t = 0; t < nt; ++t)
        for (int j = 0; j < 3; ++j) {
            const uint64_t p = tri_points[3 * t + j], q = tri_points[3 * t + (j + 1) % 3];
            if (p != q) edges.push_back({std::min(p, q), std::max(p, q), (uint32_t)t});
        }
    std::sort(edges.begin(), edges.end(), [](const Edge &x, const Edge &y) { return x.a != y.a ? x.a < y.a : x.b != y.b ? x.b < y.b : x.t < y.t; });
    auto edge_neighbours = [&](uint32_t t, uint64_t p, uint64_t q, std::vector<uint32_t> &out) {
        out.clear();
        if (p == q) return;
        const Edge key{std::min(p, q), std::max(p, q), 0u};
        auto it = std::lower_bound(edges.begin(), edges.end(), key, [](const Edge &x, const Edge &y) { return x.a != y.a ? x.a < y.a : x.b < y.b; });
        for (; it != edges.end() && it->a == key.a && it->b == key.b; ++it)
            if (it->t != t && (out.empty() || out.back() != it->t)) out.push_back(it->t);      // (a triangle with a repeated edge counts once)
    };
    std::vector<uint32_t> order(np);
    for (size_t i = 0; i < np; ++i) order[i] = (uint32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return pts[3 * (size_t)a] < pts[3 * (size_t)b]; });
    std::vector<uint32_t> wave, wave2, nei;
    for (uint32_t pid : order) {
        double best = 0.0; int64_t seed = -1; bool reverse = false;
        for (uint32_t c : cells_of[pid]) {
            if (visited[c]) continue;
            const double nx = unit_normal_x(c);
            if (std::fabs(nx) > best) { best = std::fabs(nx); seed = c; reverse = nx > 0.0; }
        }
        if (seed < 0) continue;
        if (reverse) reverse_cell((size_t)seed);
        visited[(size_t)seed] = 1;
        wave.assign(1, (uint32_t)seed);
        while (!wave.empty()) {
            wave2.clear();
            for (uint32_t c : wave)
                for (int j = 0; j < 3; ++j) {
                    const uint64_t p1 = cur[3 * (size_t)c + j], p2 = cur[3 * (size_t)c + (j + 1) % 3];
                    edge_neighbours(c, p1, p2, nei);
                    if (nei.size() != 1) continue;                 // boundary or non-manifold edge: orientation does not cross it
                    const uint32_t nb = nei[0];
                    if (visited[nb]) continue;
                    int l = 0;
                    while (l < 3 && cur[3 * (size_t)nb + l] != p2) ++l;
                    if (l < 3 && cur[3 * (size_t)nb + (l + 1) % 3] != p1) reverse_cell(nb);     // it walks the edge p1 -> p2 too: inconsistent
                    visited[nb] = 1;
                    wave2.push_back(nb);
                }
            wave.swap(wave2);
        }
    }
    return flip;
}

// ---- colour ramps, include/Util/ColorRamp.cuh:31-112 ----
struct Stop { float position; float rgb[3]; };
const std::vector<Stop> &stops_for(const std::string &preset_any_case) {
    static const std::vector<Stop> viridis = {{0.00f, {0.267f, 0.004f, 0.329f}}, {0.25f, {0.283f, 0.141f, 0.458f}}, {0.50f, {0.254f, 0.265f, 0.530f}},
                                              {0.75f, {0.196f, 0.509f, 0.364f}}, {1.00f, {0.993f, 0.906f, 0.144f}}};
    static const std::vector<Stop> plasma = {{0.00f, {0.050f, 0.030f, 0.527f}}, {0.25f, {0.537f, 0.062f, 0.549f}}, {0.50f, {0.871f, 0.191f, 0.494f}},
                                             {0.75f, {0.992f, 0.580f, 0.288f}}, {1.00f, {0.940f, 0.975f, 0.131f}}};
    static const std::vector<Stop> spectral = {{0.00f, {0.619f, 0.003f, 0.258f}}, {0.20f, {0.835f, 0.243f, 0.310f}}, {0.40f, {0.957f, 0.427f, 0.263f}},
                                               {0.60f, {0.993f, 0.681f, 0.380f}}, {0.80f, {0.741f, 0.858f, 0.407f}}, {1.00f, {0.400f, 0.761f, 0.647f}}};
    static const std::vector<Stop> terrain = {{0.00f, {0.149f, 0.149f, 0.149f}}, {0.25f, {0.114f, 0.451f, 0.208f}}, {0.50f, {0.639f, 0.784f, 0.325f}},
                                              {0.75f, {0.988f, 0.972f, 0.745f}}, {1.00f, {0.996f, 0.922f, 0.545f}}};
    static const std::vector<Stop> heatmap = {{0.00f, {0.050f, 0.050f, 0.300f}}, {0.25f, {0.000f, 0.000f, 1.000f}}, {0.50f, {0.000f, 1.000f, 1.000f}},
                                              {0.75f, {1.000f, 1.000f, 0.000f}}, {1.00f, {1.000f, 0.000f, 0.000f}}};
    static const std::vector<Stop> grayscale = {{0.00f, {0.050f, 0.050f, 0.050f}}, {1.00f, {0.950f, 0.950f, 0.950f}}};
    const std::string p = lower(preset_any_case);
    if (p == "plasma") return plasma;
    if (p == "spectral") return spectral;
    if (p == "terrain") return terrain;
    if (p == "heatmap") return heatmap;
    if (p == "grayscale") return grayscale;
    return viridis;                                      // also the fallback for unknown names
}

}  // namespace

extern "C" {

const char *hrt_io_last_error(void) { return g_error.c_str(); }

// ---- ASCII STL --------------------------------------------------------------------------------------
int hrt_io_read_stl(const char *path, HrtIoMesh *out) {
    if (!path || !out) return fail("hrt_io_read_stl: NULL argument");
    std::memset(out, 0, sizeof *out);
    std::string text;
    if (!read_text(path, text)) return fail(std::string("cannot open STL file ") + path);
    Tokens tk(text);
    std::string t;
    if (!tk.next(t) || t != "solid") return fail(std::string(path) + ": not an ASCII STL file (binary STL is not produced by the reference's data source)");
    tk.skip_line();
    std::vector<float> verts, normals, fnormals;
    while (tk.next(t)) {
        if (t == "endsolid") break;
        if (t != "facet") return fail(std::string(path) + ": 'facet' expected, got '" + t + "'");
        double n[3], v[3][3];
        if (!tk.next(t) || t != "normal" || !tk.number(n[0]) || !tk.number(n[1]) || !tk.number(n[2])) return fail(std::string(path) + ": bad facet normal");
        if (!tk.next(t) || t != "outer" || !tk.next(t) || t != "loop") return fail(std::string(path) + ": 'outer loop' expected");
        for (int k = 0; k < 3; ++k)
            if (!tk.next(t) || t != "vertex" || !tk.number(v[k][0]) || !tk.number(v[k][1]) || !tk.number(v[k][2])) return fail(std::string(path) + ": bad vertex");
        if (!tk.next(t) || t != "endloop" || !tk.next(t) || t != "endfacet") return fail(std::string(path) + ": 'endloop endfacet' expected");
        // vtkSTLReader keeps single-precision points; the reference then casts the double it gets back to float
        float fv[3][3];
        for (int k = 0; k < 3; ++k) for (int a = 0; a < 3; ++a) { fv[k][a] = (float)v[k][a]; verts.push_back(fv[k][a]); }
        // cell normal as vtkPolyDataNormals computes it: normalised cross product of the winding, in double
        const double e1[3] = {(double)fv[1][0] - fv[0][0], (double)fv[1][1] - fv[0][1], (double)fv[1][2] - fv[0][2]};
        const double e2[3] = {(double)fv[2][0] - fv[0][0], (double)fv[2][1] - fv[0][1], (double)fv[2][2] - fv[0][2]};
        double c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        const double len = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
        if (len > 0.0) { c[0] /= len; c[1] /= len; c[2] /= len; }
        for (int k = 0; k < 3; ++k) for (int a = 0; a < 3; ++a) normals.push_back((float)c[a]);
        for (int a = 0; a < 3; ++a) fnormals.push_back((float)n[a]);
    }
    {   // vtkSTLReader merges coincident points (Merging is on by default: exact equality of the float coordinates), which is what
        // gives vtkPolyDataNormals the connectivity for its consistency / auto-orientation pass; a reversed facet's normal is negated
        std::map<std::array<uint32_t, 3>, uint64_t> ids;
        std::vector<uint64_t> tri_points; std::vector<double> pts;
        for (size_t v = 0; v < verts.size() / 3; ++v) {
            std::array<uint32_t, 3> key;
            for (int a = 0; a < 3; ++a) { float f = verts[3 * v + a]; if (f == 0.0f) f = 0.0f; std::memcpy(&key[a], &f, 4); }      // (-0 == +0)
            auto it = ids.find(key);
            if (it == ids.end()) { it = ids.emplace(key, (uint64_t)(pts.size() / 3)).first; for (int a = 0; a < 3; ++a) pts.push_back((double)verts[3 * v + a]); }
            tri_points.push_back(it->second);
        }
        const std::vector<uint8_t> flip = orient_triangles(tri_points, pts);
        for (size_t t2 = 0; t2 < flip.size(); ++t2)
            if (flip[t2]) for (int k = 0; k < 9; ++k) normals[9 * t2 + k] = -normals[9 * t2 + k];
    }
    out->n_triangles = verts.size() / 9;
    out->vertices = dup_array(verts); out->normals = dup_array(normals); out->file_normals = dup_array(fnormals);
    if (!out->vertices || !out->normals || !out->file_normals) { hrt_io_free_mesh(out); return fail("out of memory"); }
    return 0;
}

void hrt_io_free_mesh(HrtIoMesh *m) {
    if (!m) return;
    std::free(m->vertices); std::free(m->normals); std::free(m->file_normals);
    std::memset(m, 0, sizeof *m);
}

// ---- legacy ASCII VTK POLYDATA with POINT_DATA ------------------------------------------------------------
int hrt_io_read_particle_vtk(const char *path, HrtIoParticles *out) {
    if (!path || !out) return fail("hrt_io_read_particle_vtk: NULL argument");
    std::memset(out, 0, sizeof *out);
    std::string text;
    if (!read_text(path, text)) return fail(std::string("cannot open VTK file ") + path);
    if (text.compare(0, 14, "# vtk DataFile") != 0) return fail(std::string(path) + ": not a legacy VTK file");
    Tokens tk(text);
    tk.skip_line(); ++tk.p;                             // version line
    tk.skip_line(); ++tk.p;                             // title line
    std::string t;
    if (!tk.next(t) || t != "ASCII") return fail(std::string(path) + ": only ASCII legacy VTK is supported");
    uint64_t n = 0;
    std::vector<double> points, quat, vel, id, shape;
    bool have_points = false, in_point_data = false;
    auto read_values = [&](size_t count, std::vector<double> *dst) -> bool {
        for (size_t i = 0; i < count; ++i) { double d; if (!tk.number(d)) return false; if (dst) dst->push_back(d); }
        return true;
    };
    while (tk.next(t)) {
        if (t == "DATASET") { if (!tk.next(t) || t != "POLYDATA") return fail(std::string(path) + ": DATASET POLYDATA expected"); }
        else if (t == "POINTS") {
            double cnt; std::string type;
            if (!tk.number(cnt) || !tk.next(type)) return fail(std::string(path) + ": bad POINTS header");
            n = (uint64_t)cnt;
            if (!read_values(3 * n, &points)) return fail(std::string(path) + ": short POINTS block");
            have_points = true;
        } else if (t == "VERTICES" || t == "LINES" || t == "POLYGONS" || t == "TRIANGLE_STRIPS") {
            double cells, size;
            if (!tk.number(cells) || !tk.number(size) || !read_values((size_t)size, nullptr)) return fail(std::string(path) + ": bad " + t + " block");
        } else if (t == "POINT_DATA") {
            double cnt;
            if (!tk.number(cnt) || (uint64_t)cnt != n) return fail(std::string(path) + ": POINT_DATA count differs from POINTS");
            in_point_data = true;
        } else if (t == "CELL_DATA") {
            double cnt; if (!tk.number(cnt)) return fail(std::string(path) + ": bad CELL_DATA"); in_point_data = false;
        } else if (t == "SCALARS") {
            std::string name, type;
            if (!tk.next(name) || !tk.next(type)) return fail(std::string(path) + ": bad SCALARS header");
            // optional component count, then LOOKUP_TABLE <name>
            size_t comps = 1;
            const size_t save = tk.p;
            std::string maybe;
            if (!tk.next(maybe)) return fail(std::string(path) + ": truncated SCALARS");
            if (maybe != "LOOKUP_TABLE") { comps = (size_t)std::strtoul(maybe.c_str(), nullptr, 10); if (comps == 0) { tk.p = save; comps = 1; } else if (!tk.next(maybe)) return fail(std::string(path) + ": truncated SCALARS"); }
            if (maybe != "LOOKUP_TABLE" || !tk.next(maybe)) return fail(std::string(path) + ": LOOKUP_TABLE expected after SCALARS " + name);
            std::vector<double> *dst = nullptr;
            if (in_point_data && name == "quat" && comps == 4) dst = &quat;
            else if (in_point_data && name == "id") dst = &id;
            else if (in_point_data && name == "shape_id") dst = &shape;
            if (!read_values(comps * n, dst)) return fail(std::string(path) + ": short SCALARS " + name);
        } else if (t == "VECTORS" || t == "NORMALS") {
            std::string name, type;
            if (!tk.next(name) || !tk.next(type)) return fail(std::string(path) + ": bad " + t + " header");
            if (!read_values(3 * n, (in_point_data && t == "VECTORS" && name == "vel") ? &vel : nullptr)) return fail(std::string(path) + ": short " + t + " " + name);
        } else if (t == "FIELD" || t == "METADATA") {
            if (!skip_vtk_block(tk, t)) return fail(std::string(path) + ": malformed " + t + " block");
        } else {
            return fail(std::string(path) + ": unsupported keyword '" + t + "'");
        }
    }
    if (!have_points) return fail(std::string(path) + ": no POINTS");
    if (id.size() != n || shape.size() != n || quat.size() != 4 * n || vel.size() != 3 * n)
        return fail(std::string(path) + ": POINT_DATA must hold id, shape_id, quat (4 components) and vel (RendererTime needs all four)");
    std::vector<HrtParticleState> st(n);
    std::vector<uint64_t> ids(n), shapes(n);
    for (uint64_t i = 0; i < n; ++i) {
        std::memset(&st[i], 0, sizeof st[i]);
        st[i].quat = HrtFloat4{(float)quat[4 * i], (float)quat[4 * i + 1], (float)quat[4 * i + 2], (float)quat[4 * i + 3]};
        st[i].position = HrtFloat3{(float)points[3 * i], (float)points[3 * i + 1], (float)points[3 * i + 2]};
        st[i].velocity = HrtFloat3{(float)vel[3 * i], (float)vel[3 * i + 1], (float)vel[3 * i + 2]};
        ids[i] = (uint64_t)id[i]; shapes[i] = (uint64_t)shape[i];
    }
    out->n = n; out->states = dup_array(st); out->ids = dup_array(ids); out->shape_ids = dup_array(shapes);
    if (!out->states || !out->ids || !out->shape_ids) { hrt_io_free_particles(out); return fail("out of memory"); }
    return 0;
}

void hrt_io_free_particles(HrtIoParticles *p) {
    if (!p) return;
    std::free(p->states); std::free(p->ids); std::free(p->shape_ids);
    std::memset(p, 0, sizeof *p);
}

// ---- *.vtk.series -------------------------------------------------------------------------------------
int hrt_io_read_series(const char *directory, const char *name, HrtIoSeries *out) {
    if (!directory || !name || !out) return fail("hrt_io_read_series: NULL argument");
    std::memset(out, 0, sizeof *out);
    const std::string dir = directory, path = dir + name;
    std::string text;
    if (!read_text(path, text)) return fail("cannot open series file " + path);
    try {
        const hrt_io::Json data = hrt_io::JsonParser(text).parse();
        (void)data.at("file-series-version").string();
        if (!data.contains("files") || data.at("files").kind != hrt_io::Json::Array) return fail(path + ": no \"files\" array");
        std::vector<std::string> files; std::vector<float> times;
        for (const auto &item : data.at("files").arr) { files.push_back(dir + item.at("name").string()); times.push_back(item.at("time").number_f()); }
        const size_t n = files.size();
        std::vector<float> dur(n);
        if (n == 1) dur[0] = 1000.0f;
        else if (n > 1) { for (size_t i = 0; i + 1 < n; ++i) dur[i] = times[i + 1] - times[i]; dur[n - 1] = dur[n - 2]; }
        out->n = n;
        out->durations = dup_array(dur);
        out->files = static_cast<char **>(std::calloc(std::max<size_t>(1, n), sizeof(char *)));
        if (!out->durations || !out->files) { hrt_io_free_series(out); return fail("out of memory"); }
        for (size_t i = 0; i < n; ++i) out->files[i] = dup_string(files[i]);
    } catch (const std::exception &e) { hrt_io_free_series(out); return fail(path + ": " + e.what()); }
    return 0;
}

void hrt_io_free_series(HrtIoSeries *s) {
    if (!s) return;
    if (s->files) for (uint64_t i = 0; i < s->n; ++i) std::free(s->files[i]);
    std::free(s->files); std::free(s->durations);
    std::memset(s, 0, sizeof *s);
}

// ---- colour ramp ---------------------------------------------------------------------------------------
int hrt_io_bake_color_ramp(const char *preset, uint64_t count, float *out_rgb) {
    if (!preset || (count && !out_rgb)) return fail("hrt_io_bake_color_ramp: NULL argument");
    const std::vector<Stop> &stops = stops_for(preset);
    if (count == 0) return 0;
    if (count == 1) { std::memcpy(out_rgb, stops.back().rgb, 12); return 0; }
    for (uint64_t i = 0; i < count; ++i) {
        const float u = (float)i / (float)(count - 1);
        const Stop *lo = &stops.front(), *hi = &stops.back();
        for (size_t s = 1; s < stops.size(); ++s) {
            if (u <= stops[s].position) { hi = &stops[s]; lo = &stops[s - 1]; break; }
            lo = &stops[s];
        }
        const float span = hi->position - lo->position;
        float t = span > 0.0f ? (u - lo->position) / span : 0.0f;
        t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
        for (int a = 0; a < 3; ++a) out_rgb[3 * i + a] = lo->rgb[a] + (hi->rgb[a] - lo->rgb[a]) * t;
    }
    return 0;
}

int hrt_io_construct_transform(const float *shift3, const float *rotate_deg3, const float *scale3, float *out12) {
    if (!shift3 || !rotate_deg3 || !scale3 || !out12) return fail("hrt_io_construct_transform: NULL argument");
    construct_transform(shift3, rotate_deg3, scale3, out12);
    return 0;
}

// ---- config.json ---------------------------------------------------------------------------------------
int hrt_io_load_config(const char *path, HrtIoConfig *out) {
    if (!path || !out) return fail("hrt_io_load_config: NULL argument");
    std::memset(out, 0, sizeof *out);
    std::string text;
    if (!read_text(path, text)) return fail(std::string("Failed to open config: ") + path);
    try {
        const hrt_io::Json data = hrt_io::JsonParser(text).parse();
        std::vector<float> roughs, metals;
        for (const auto &r : data.at("roughs").arr) { const auto a = r.at("albedo").floats(3); roughs.insert(roughs.end(), a.begin(), a.begin() + 3); }
        for (const auto &m : data.at("metals").arr) { const auto a = m.at("albedo").floats(3); metals.insert(metals.end(), a.begin(), a.begin() + 3); metals.push_back(m.at("fuzz").number_f()); }
        std::vector<HrtIoSphere> spheres;
        for (const auto &s : data.at("spheres").arr) {
            HrtIoSphere sp; std::memset(&sp, 0, sizeof sp);
            const auto c = s.at("center").floats(3), sh = s.at("shift").floats(3), ro = s.at("rotate").floats(3), sc = s.at("scale").floats(3);
            std::memcpy(sp.center, c.data(), 12); sp.radius = s.at("radius").number_f();
            sp.metal = s.at("mat-type").string() == "ROUGH" ? 0 : 1;            // anything but "ROUGH" is METAL, as in the reference
            sp.material_index = (uint64_t)s.at("mat-index").number();
            construct_transform(sh.data(), ro.data(), sc.data(), sp.transform);
            spheres.push_back(sp);
        }
        const hrt_io::Json &loop = data.at("loop-data");
        const std::string api = loop.at("api").string();
        if (api == "D3D11" || api == "D3D12") return fail("Direct3D (D3D11/D3D12) is only supported on Windows; use \"OGL\" or \"VK\"");
        if (api != "OGL" && api != "VK") return fail("Invalid api type, must be \"OGL\", \"VK\", \"D3D11\" or \"D3D12\"");
        out->api_is_opengl = api == "OGL";
        out->mesh = data.at("mesh").boolean(); out->cache = data.at("cache").boolean(); out->debug_mode = data.at("debug-mode").boolean();
        out->cache_process_thread_count = (uint64_t)data.at("cache-process-thread-count").number();
        out->window_width = (int32_t)loop.at("window-width").number(); out->window_height = (int32_t)loop.at("window-height").number();
        out->fps = (uint64_t)loop.at("fps").number(); out->render_speed_ratio = (uint64_t)loop.at("render-speed-ratio").number();
        out->camera_initial_speed_ratio = (uint64_t)loop.at("camera-initial-speed-ratio").number();
        const auto cc = loop.at("camera-center").floats(3), ct = loop.at("camera-target").floats(3), up = loop.at("up-direction").floats(3);
        const auto ps = loop.at("particle-shift").floats(3), psc = loop.at("particle-scale").floats(3);
        std::memcpy(out->camera_center, cc.data(), 12); std::memcpy(out->camera_target, ct.data(), 12); std::memcpy(out->up_direction, up.data(), 12);
        std::memcpy(out->particle_shift, ps.data(), 12); std::memcpy(out->particle_scale, psc.data(), 12);
        out->mouse_sensitivity = loop.at("mouse-sensitivity").number_f();
        out->camera_pitch_limit_degree = loop.at("camera-pitch-limit-degree").number_f();
        out->camera_speed_stride = loop.at("camera-speed-stride").number_f();
        out->series_path = dup_string(data.at("series-path").string()); out->series_name = dup_string(data.at("series-name").string());
        out->cache_path = dup_string(data.at("cache-path").string()); out->stl_path = dup_string(data.at("stl-path").string());
        out->particle_material_preset = dup_string(data.at("particle-material-preset").string()); out->api = dup_string(api);
        out->roughs = dup_array(roughs); out->n_roughs = roughs.size() / 3;
        out->metals = dup_array(metals); out->n_metals = metals.size() / 4;
        out->spheres = dup_array(spheres); out->n_spheres = spheres.size();
    } catch (const std::exception &e) { hrt_io_free_config(out); return fail(std::string(path) + ": " + e.what()); }
    return 0;
}

void hrt_io_free_config(HrtIoConfig *c) {
    if (!c) return;
    std::free(c->series_path); std::free(c->series_name); std::free(c->cache_path); std::free(c->stl_path);
    std::free(c->particle_material_preset); std::free(c->api);
    std::free(c->roughs); std::free(c->metals); std::free(c->spheres);
    std::memset(c, 0, sizeof *c);
}

// ---- Mesh-mode cache: [u64 particles] { [u64 id] [float3 velocity] [u64 vertices] [float3 x N] [float3 x N] } ----
int hrt_io_read_mesh_cache(const char *path, HrtIoMeshCache *out) {
    if (!path || !out) return fail("hrt_io_read_mesh_cache: NULL argument");
    std::memset(out, 0, sizeof *out);
    std::ifstream in(path, std::ios::in | std::ios::binary);
    if (!in) return fail(std::string("cannot open cache file ") + path);
    uint64_t count = 0;
    in.read(reinterpret_cast<char *>(&count), 8);
    if (!in) return fail(std::string(path) + ": truncated header");
    std::vector<uint64_t> ids, first{0};
    std::vector<float> vel, verts, normals;
    for (uint64_t i = 0; i < count; ++i) {
        uint64_t id = 0, nv = 0; float v[3];
        in.read(reinterpret_cast<char *>(&id), 8); in.read(reinterpret_cast<char *>(v), 12); in.read(reinterpret_cast<char *>(&nv), 8);
        if (!in) return fail(std::string(path) + ": truncated particle header");
        if (nv % 3 != 0 || nv > (1ull << 40)) return fail(std::string(path) + ": vertex count is not a multiple of 3");
        const size_t at = verts.size();
        verts.resize(at + 3 * nv); normals.resize(at + 3 * nv);
        in.read(reinterpret_cast<char *>(verts.data() + at), (std::streamsize)(12 * nv));
        in.read(reinterpret_cast<char *>(normals.data() + at), (std::streamsize)(12 * nv));
        if (!in) return fail(std::string(path) + ": truncated vertex / normal arrays");
        ids.push_back(id); vel.insert(vel.end(), v, v + 3); first.push_back(first.back() + nv / 3);
    }
    out->n_particles = count; out->ids = dup_array(ids); out->velocities = dup_array(vel); out->first_triangle = dup_array(first);
    out->vertices = dup_array(verts); out->normals = dup_array(normals);
    if (!out->ids || !out->velocities || !out->first_triangle || !out->vertices || !out->normals) { hrt_io_free_mesh_cache(out); return fail("out of memory"); }
    return 0;
}

int hrt_io_write_mesh_cache(const char *path, const HrtIoMeshCache *c) {
    if (!path || !c) return fail("hrt_io_write_mesh_cache: NULL argument");
    std::ofstream o(path, std::ios::out | std::ios::binary);
    if (!o) return fail(std::string("cannot create cache file ") + path);
    o.write(reinterpret_cast<const char *>(&c->n_particles), 8);
    for (uint64_t i = 0; i < c->n_particles; ++i) {
        const uint64_t t0 = c->first_triangle[i], nv = 3 * (c->first_triangle[i + 1] - t0);
        o.write(reinterpret_cast<const char *>(&c->ids[i]), 8);
        o.write(reinterpret_cast<const char *>(c->velocities + 3 * i), 12);
        o.write(reinterpret_cast<const char *>(&nv), 8);
        o.write(reinterpret_cast<const char *>(c->vertices + 9 * t0), (std::streamsize)(12 * nv));
        o.write(reinterpret_cast<const char *>(c->normals + 9 * t0), (std::streamsize)(12 * nv));
    }
    o.close();
    return o ? 0 : fail(std::string("write failed: ") + path);
}

// ---- metadata.cache -----------------------------------------------------------------------------------
int hrt_io_read_metadata_cache(const char *directory, uint64_t *out_max_cell_count) {
    if (!directory || !out_max_cell_count) return fail("hrt_io_read_metadata_cache: NULL argument");
    const std::string path = std::string(directory) + "metadata.cache";
    std::ifstream in(path, std::ios::in);
    if (!in) return fail("cannot open " + path);
    uint64_t v = 0;
    in >> v;                                             // `metaData >> maxCount`, VTKMeshReader.cu:277
    if (in.fail()) return fail(path + ": no cell count");
    *out_max_cell_count = v;
    return 0;
}

int hrt_io_write_metadata_cache(const char *directory, uint64_t max_cell_count) {
    if (!directory) return fail("hrt_io_write_metadata_cache: NULL argument");
    const std::string path = std::string(directory) + "metadata.cache";
    std::ofstream o(path, std::ios::out);
    if (!o) return fail("cannot create " + path);
    o << max_cell_count;                                 // `metaData << maxCellCount`, VTKMeshReader.cu:203: decimal text, no newline
    o.close();
    return o ? 0 : fail("write failed: " + path);
}

// ---- Mesh-mode VTK file: triangle strips + cell data ---------------------------------------------------
int hrt_io_read_vtk_mesh_file(const char *path, HrtIoMeshCache *out, uint64_t *out_cell_count) {
    if (!path || !out) return fail("hrt_io_read_vtk_mesh_file: NULL argument");
    std::memset(out, 0, sizeof *out);
    std::string text;
    if (!read_text(path, text)) return fail(std::string("cannot open VTK file ") + path);
    if (text.compare(0, 14, "# vtk DataFile") != 0) return fail(std::string(path) + ": not a legacy VTK file");
    Tokens tk(text);
    tk.skip_line(); ++tk.p;                             // version line
    tk.skip_line(); ++tk.p;                             // title line
    std::string t;
    if (!tk.next(t) || t != "ASCII") return fail(std::string(path) + ": only ASCII legacy VTK is supported");
    std::vector<double> points, id, vel;
    std::vector<std::vector<uint64_t>> strips;
    uint64_t n_points = 0, n_cells = 0; bool in_cell_data = false;
    auto read_values = [&](size_t count, std::vector<double> *dst) -> bool {
        for (size_t i = 0; i < count; ++i) { double d; if (!tk.number(d)) return false; if (dst) dst->push_back(d); }
        return true;
    };
    while (tk.next(t)) {
        if (t == "DATASET") { if (!tk.next(t) || t != "POLYDATA") return fail(std::string(path) + ": DATASET POLYDATA expected"); }
        else if (t == "POINTS") {
            double cnt; std::string type;
            if (!tk.number(cnt) || !tk.next(type)) return fail(std::string(path) + ": bad POINTS header");
            n_points = (uint64_t)cnt;
            if (!read_values(3 * n_points, &points)) return fail(std::string(path) + ": short POINTS block");
        } else if (t == "TRIANGLE_STRIPS") {
            double cells, size;
            if (!tk.number(cells) || !tk.number(size)) return fail(std::string(path) + ": bad TRIANGLE_STRIPS header");
            for (uint64_t c = 0; c < (uint64_t)cells; ++c) {
                double k;
                if (!tk.number(k) || k < 3) return fail(std::string(path) + ": a triangle strip needs at least 3 points");
                std::vector<double> idx;
                if (!read_values((size_t)k, &idx)) return fail(std::string(path) + ": short TRIANGLE_STRIPS block");
                std::vector<uint64_t> s;
                for (double d : idx) { if (d < 0 || (uint64_t)d >= n_points) return fail(std::string(path) + ": strip point index out of range"); s.push_back((uint64_t)d); }
                strips.push_back(std::move(s));
            }
        } else if (t == "VERTICES" || t == "LINES" || t == "POLYGONS") {
            // the reference rejects any cell that is not a vtkTriangleStrip (VTKReaderImpl.cpp:73-77)
            double cells, size;
            if (!tk.number(cells) || !tk.number(size)) return fail(std::string(path) + ": bad " + t + " block");
            if (cells > 0) return fail(std::string(path) + ": found illegal cell type (" + t + "): Mesh mode takes triangle strips only");
        } else if (t == "CELL_DATA") {
            double cnt; if (!tk.number(cnt)) return fail(std::string(path) + ": bad CELL_DATA");
            n_cells = (uint64_t)cnt; in_cell_data = true;
        } else if (t == "POINT_DATA") {
            double cnt; if (!tk.number(cnt)) return fail(std::string(path) + ": bad POINT_DATA"); in_cell_data = false; n_cells = (uint64_t)cnt;
        } else if (t == "SCALARS") {
            std::string name, type, maybe;
            if (!tk.next(name) || !tk.next(type)) return fail(std::string(path) + ": bad SCALARS header");
            size_t comps = 1;
            const size_t save = tk.p;
            if (!tk.next(maybe)) return fail(std::string(path) + ": truncated SCALARS");
            if (maybe != "LOOKUP_TABLE") { comps = (size_t)std::strtoul(maybe.c_str(), nullptr, 10); if (comps == 0) { tk.p = save; comps = 1; } else if (!tk.next(maybe)) return fail(std::string(path) + ": truncated SCALARS"); }
            if (maybe != "LOOKUP_TABLE" || !tk.next(maybe)) return fail(std::string(path) + ": LOOKUP_TABLE expected after SCALARS " + name);
            if (!read_values(comps * n_cells, (in_cell_data && name == "id" && comps == 1) ? &id : nullptr)) return fail(std::string(path) + ": short SCALARS " + name);
        } else if (t == "VECTORS" || t == "NORMALS") {
            std::string name, type;
            if (!tk.next(name) || !tk.next(type)) return fail(std::string(path) + ": bad " + t + " header");
            if (!read_values(3 * n_cells, (in_cell_data && t == "VECTORS" && name == "vel") ? &vel : nullptr)) return fail(std::string(path) + ": short " + t + " " + name);
        } else if (t == "FIELD" || t == "METADATA") {
            if (!skip_vtk_block(tk, t)) return fail(std::string(path) + ": malformed " + t + " block");
        } else {
            return fail(std::string(path) + ": unsupported keyword '" + t + "'");
        }
    }
    if (n_points == 0) return fail(std::string(path) + ": failed to get poly data or there is no points in file");     // VTKReaderImpl.cpp:38-41
    if (id.size() != strips.size() || vel.size() != 3 * strips.size()) return fail(std::string(path) + ": failed to read cell data (id and vel per cell)");   // :49-52
    // triangles of the strips (:92-104) and the unit normal of each; point normals = normalised sum over the triangles using the point
    std::vector<uint64_t> ids, first{0};
    std::vector<float> velocities, verts, normals;
    std::vector<double> pn(3 * n_points, 0.0);
    struct Tri { uint64_t a, b, c; };
    std::vector<std::vector<Tri>> tris(strips.size());
    std::vector<uint64_t> tri_points;
    for (size_t s2 = 0; s2 < strips.size(); ++s2) {
        const auto &st = strips[s2];
        for (size_t k = 0; k + 2 < st.size(); ++k) {
            Tri tr{st[k], st[k + 1], st[k + 2]};
            if (k & 1) std::swap(tr.b, tr.c);
            tris[s2].push_back(tr);
            tri_points.push_back(tr.a); tri_points.push_back(tr.b); tri_points.push_back(tr.c);
        }
    }
    // vtkPolyDataNormals works on the strips' triangles, makes their order consistent across shared edges and orients each
    // connected component outwards (orient_triangles above); the point normals are sums over the triangles SO ORIENTED
    const std::vector<uint8_t> flip = orient_triangles(tri_points, points);
    {
        size_t ti = 0;
        for (size_t s2 = 0; s2 < strips.size(); ++s2)
            for (const Tri &tr : tris[s2]) {
                const double *A = &points[3 * tr.a], *B = &points[3 * tr.b], *C = &points[3 * tr.c];
                const double e1[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, e2[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
                double nrm[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                const double len = std::sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
                const double sign = flip[ti++] ? -1.0 : 1.0;
                if (len > 0.0) for (uint64_t v : {tr.a, tr.b, tr.c}) for (int d = 0; d < 3; ++d) pn[3 * v + d] += sign * nrm[d] / len;
            }
    }
    for (uint64_t v = 0; v < n_points; ++v) {
        const double len = std::sqrt(pn[3 * v] * pn[3 * v] + pn[3 * v + 1] * pn[3 * v + 1] + pn[3 * v + 2] * pn[3 * v + 2]);
        if (len > 0.0) for (int d = 0; d < 3; ++d) pn[3 * v + d] /= len;
    }
    for (size_t s2 = 0; s2 < strips.size(); ++s2) {
        ids.push_back((uint64_t)id[s2]);
        for (int d = 0; d < 3; ++d) velocities.push_back((float)vel[3 * s2 + d]);
        for (const Tri &tr : tris[s2])
            for (uint64_t v : {tr.a, tr.b, tr.c})
                for (int d = 0; d < 3; ++d) { verts.push_back((float)points[3 * v + d]); normals.push_back((float)pn[3 * v + d]); }
        first.push_back(first.back() + tris[s2].size());
    }
    out->n_particles = strips.size(); out->ids = dup_array(ids); out->velocities = dup_array(velocities); out->first_triangle = dup_array(first);
    out->vertices = dup_array(verts); out->normals = dup_array(normals);
    if (!out->ids || !out->velocities || !out->first_triangle || !out->vertices || !out->normals) { hrt_io_free_mesh_cache(out); return fail("out of memory"); }
    if (out_cell_count) *out_cell_count = strips.size();
    return 0;
}

void hrt_io_free_mesh_cache(HrtIoMeshCache *c) {
    if (!c) return;
    std::free(c->ids); std::free(c->velocities); std::free(c->first_triangle); std::free(c->vertices); std::free(c->normals);
    std::memset(c, 0, sizeof *c);
}

}  // extern "C"
