// ff_host.cpp — host-side pieces of the ABI that need no GPU: error strings, the reference's struct
// constructors (Geometry, Camera, BXDF) restated with glm's operation order, and the OBJ reader.
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "ff_internal.h"
#include "ff_math.h"

namespace ff {

static thread_local std::string g_last_error;

int fail(int status, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return status;
}

void clear_error() { g_last_error.clear(); }

} // namespace ff

using namespace ffm;

extern "C" {

const char* ff_last_error(void) { return ff::g_last_error.c_str(); }

int ff_version(void) { return 100; /* 0.1.0 */ }

void ff_bxdf_init(FfBXDF* b)
{
    // utilities.h:81-88
    if (!b) return;
    b->m_type = FF_BXDF_COUNT;
    b->m_albedo = FfVec3{ -1, -1, -1 };
    b->m_specularColor = FfVec3{ -1, -1, -1 };
    b->m_refractiveIndex = -1;
    b->m_emissiveColor = FfVec3{ -1, -1, -1 };
    b->m_intensity = -1;
    b->m_transmittanceColor = FfVec3{ -1, -1, -1 };
}

void ff_geometry_init(FfGeometry* g, int geometry_type, FfVec3 position, FfVec3 rotation_deg, FfVec3 scale_v,
                      FfTriangle* triangles, int number_of_triangles, float radius)
{
    // Geometry::Geometry, utilities.h:176-213
    if (!g) return;
    std::memset(g, 0, sizeof *g);
    g->m_geometryType = geometry_type;
    g->m_position = position;
    g->m_rotation = rotation_deg;
    g->m_scale = scale_v;
    g->m_normal = FfVec3{ 0.f, 0.f, 1.f }; // utilities.h:229

    const M4 I = identity();
    const M4 T = translate(I, v3(position.x, position.y, position.z));                 // :180
    M4 R = rotate(I, radians(rotation_deg.x), v3(1.f, 0.f, 0.f));                      // :182
    R = mul(R, rotate(I, radians(rotation_deg.y), v3(0.f, 1.f, 0.f)));                 // :183
    R = mul(R, rotate(I, radians(rotation_deg.z), v3(0.f, 0.f, 1.f)));                 // :184
    const M4 S = scale(I, v3(scale_v.x, scale_v.y, scale_v.z));                        // :186
    const M4 model = mul(mul(T, R), S);                                                // :187
    store(model, g->m_modelMatrix.m);
    store(inverse(model), g->m_inverseModelMatrix.m);                                  // :189

    switch (geometry_type) {
    case FF_GEOM_SPHERE:
        g->m_sphereRadius = radius;
        break;
    case FF_GEOM_TRIANGLEMESH:
        if (triangles && number_of_triangles > 0) {
            g->m_numberOfTriangles = number_of_triangles;
            g->m_triangles = triangles; // borrowed; the reference copies into a new[] it never frees (:203)
        }
        break;
    default:
        break;
    }
}

void ff_camera_update_basis(FfCamera* c)
{
    // Camera::UpdateBasisAxis, utilities.h:407-418
    if (!c) return;
    const float yaw = radians(c->m_yaw), pitch = radians(c->m_pitch);
    const V3 front = v3(std::cos(yaw) * std::cos(pitch), std::sin(pitch), std::sin(yaw) * std::cos(pitch));
    const V3 fwd = normalize(front);
    const V3 up_w = v3(c->m_worldUp.x, c->m_worldUp.y, c->m_worldUp.z);
    const V3 right = normalize(cross(fwd, up_w));
    const V3 up = normalize(cross(right, fwd));
    c->m_forward = FfVec3{ fwd.x, fwd.y, fwd.z };
    c->m_right = FfVec3{ right.x, right.y, right.z };
    c->m_up = FfVec3{ up.x, up.y, up.z };
}

void ff_camera_init_default(FfCamera* c, int width, int height)
{
    if (!c) return;
    std::memset(c, 0, sizeof *c);
    c->m_cameraMovementSpeed = 0.2f;    // utilities.h:287
    c->m_cameraMouseSensitivity = 0.2f; // utilities.h:288
    c->m_position = FfVec3{ 0.f, 0.f, 15.f }; // kernel.cu:312
    c->m_forward = FfVec3{ 0.f, 0.f, -1.f };  // :313
    c->m_worldUp = FfVec3{ 0.f, 1.f, 0.f };   // :314
    c->m_fov = 70.f;                          // :315
    c->m_screenWidth = (float)width;          // :316-317 assign these swapped; un-swapped here
    c->m_screenHeight = (float)height;
    c->m_nearClip = 0.1f;                     // :318
    c->m_farClip = 1000.f;                    // :319
    c->m_pitch = 0.f;                         // :320
    c->m_yaw = -90.f;                         // :321
    ff_camera_update_basis(c);                // :322
}

void ff_camera_ray_matrix(const FfCamera* c, FfMat4* out)
{
    // kernel.cu:203 — camera.GetInverseViewMatrix() * camera.GetInverseProjectionMatrix(), utilities.h:299-317.
    // The reference recomputes this per thread; it does not depend on the pixel, so it is built once per frame here.
    if (!c || !out) return;
    const V3 pos = v3(c->m_position.x, c->m_position.y, c->m_position.z);
    const V3 fwd = v3(c->m_forward.x, c->m_forward.y, c->m_forward.z);
    const V3 up = v3(c->m_up.x, c->m_up.y, c->m_up.z);
    const M4 view = lookAtRH(pos, pos + fwd, up);
    const M4 proj = perspectiveFovRH_NO(radians(c->m_fov), c->m_screenWidth, c->m_screenHeight, c->m_nearClip, c->m_farClip);
    store(mul(inverse(view), inverse(proj)), out->m);
}

// -------------------------------------------------------------------------------------------------
// OBJ reader with LoadMesh's flattening (utilities.h:781-840).  A small from-scratch parser: v / vt / vn / f
// records, 1-based and negative (relative) indices, `v`, `v/vt`, `v//vn`, `v/vt/vn` corners.  Faces with more
// than three corners are fan-triangulated (the reference's tinyobj ear-clips; identical for convex faces, and
// every mesh the reference ships is already triangles).
// -------------------------------------------------------------------------------------------------

namespace {

struct Corner {
    int v, vt, vn;
};

bool parse_index(const char*& p, int count, int* out)
{
    char* end = nullptr;
    long i = std::strtol(p, &end, 10);
    if (end == p) return false;
    p = end;
    if (i > 0) *out = (int)i - 1;
    else if (i < 0) *out = count + (int)i;
    else *out = -1;
    return true;
}

bool parse_corner(const char*& p, int nv, int nvt, int nvn, Corner* c)
{
    c->v = c->vt = c->vn = -1;
    if (!parse_index(p, nv, &c->v)) return false;
    if (*p != '/') return true;
    ++p;
    if (*p != '/') {
        if (!parse_index(p, nvt, &c->vt)) return false;
        if (*p != '/') return true;
    }
    ++p;
    parse_index(p, nvn, &c->vn);
    return true;
}

} // namespace

int ff_load_obj(const char* path, FfTriangle** out_triangles, int* out_count)
{
    ff::clear_error();
    if (!path || !out_triangles || !out_count) return ff::fail(FF_ERR_INVALID_ARG, "ff_load_obj: null argument");
    *out_triangles = nullptr;
    *out_count = 0;
    FILE* f = std::fopen(path, "rb");
    if (!f) return ff::fail(FF_ERR_IO, "ff_load_obj: cannot open '%s': %s", path, std::strerror(errno));

    std::vector<float> pos, uv, nrm;
    std::vector<FfTriangle> tris;
    std::vector<Corner> corners;
    std::string line;
    char buf[4096];
    auto flush_line = [&](const std::string& ln) {
        const char* p = ln.c_str();
        while (*p == ' ' || *p == '\t') ++p;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            p += 2;
            for (int k = 0; k < 3; ++k) pos.push_back((float)std::strtod(p, const_cast<char**>(&p)));
        } else if (p[0] == 'v' && p[1] == 't' && (p[2] == ' ' || p[2] == '\t')) {
            p += 3;
            for (int k = 0; k < 2; ++k) uv.push_back((float)std::strtod(p, const_cast<char**>(&p)));
        } else if (p[0] == 'v' && p[1] == 'n' && (p[2] == ' ' || p[2] == '\t')) {
            p += 3;
            for (int k = 0; k < 3; ++k) nrm.push_back((float)std::strtod(p, const_cast<char**>(&p)));
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            p += 2;
            corners.clear();
            const int nv = (int)(pos.size() / 3), nvt = (int)(uv.size() / 2), nvn = (int)(nrm.size() / 3);
            for (;;) {
                while (*p == ' ' || *p == '\t') ++p;
                if (*p == '\0' || *p == '\r' || *p == '\n' || *p == '#') break;
                Corner c;
                if (!parse_corner(p, nv, nvt, nvn, &c)) break;
                corners.push_back(c);
            }
            for (size_t k = 2; k < corners.size(); ++k) {
                const Corner cs[3] = { corners[0], corners[k - 1], corners[k] };
                FfTriangle t;
                std::memset(&t, 0, sizeof t);
                FfVec3* P[3] = { &t.m_v0, &t.m_v1, &t.m_v2 };
                FfVec2* U[3] = { &t.m_uv0, &t.m_uv1, &t.m_uv2 };
                FfVec3* N[3] = { &t.m_n0, &t.m_n1, &t.m_n2 };
                bool ok = true;
                for (int q = 0; q < 3; ++q) {
                    if (cs[q].v < 0 || cs[q].v >= nv) { ok = false; break; }
                    *P[q] = FfVec3{ pos[3 * cs[q].v], pos[3 * cs[q].v + 1], pos[3 * cs[q].v + 2] };
                    if (cs[q].vt >= 0 && cs[q].vt < nvt) *U[q] = FfVec2{ uv[2 * cs[q].vt], uv[2 * cs[q].vt + 1] };
                    if (cs[q].vn >= 0 && cs[q].vn < nvn) *N[q] = FfVec3{ nrm[3 * cs[q].vn], nrm[3 * cs[q].vn + 1], nrm[3 * cs[q].vn + 2] };
                }
                if (ok) tris.push_back(t);
            }
        }
    };
    while (std::fgets(buf, sizeof buf, f)) {
        line += buf;
        if (!line.empty() && line.back() == '\n') {
            flush_line(line);
            line.clear();
        }
    }
    if (!line.empty()) flush_line(line);
    std::fclose(f);

    if (tris.empty()) return ff::fail(FF_ERR_IO, "ff_load_obj: '%s' holds no faces", path);
    FfTriangle* mem = (FfTriangle*)std::malloc(tris.size() * sizeof(FfTriangle));
    if (!mem) return ff::fail(FF_ERR_OOM, "ff_load_obj: out of memory for %zu triangles", tris.size());
    std::memcpy(mem, tris.data(), tris.size() * sizeof(FfTriangle));
    *out_triangles = mem;
    *out_count = (int)tris.size();
    return FF_OK;
}

void ff_free_triangles(FfTriangle* triangles) { std::free(triangles); }

// saveToPPM (utilities.h:842-856) for the 8-bit framebuffer: same text format (P3, "W H", 255, one "r g b" line per
// pixel, rows top to bottom as they sit in the buffer), to a caller-chosen path instead of ./render.ppm.
int ff_save_ppm(const char* path, const unsigned char* rgb8, int width, int height)
{
    ff::clear_error();
    if (!path || !rgb8 || width <= 0 || height <= 0) return ff::fail(FF_ERR_INVALID_ARG, "ff_save_ppm: bad argument");
    std::FILE* f = std::fopen(path, "w");
    if (!f) return ff::fail(FF_ERR_IO, "ff_save_ppm: cannot open %s for writing", path);
    std::fprintf(f, "P3\n%d %d\n255\n", width, height);
    const size_t n = (size_t)width * (size_t)height;
    for (size_t i = 0; i < n; ++i) std::fprintf(f, "%d %d %d\n", (int)rgb8[3 * i], (int)rgb8[3 * i + 1], (int)rgb8[3 * i + 2]);
    const bool ok = std::fflush(f) == 0 && !std::ferror(f);
    std::fclose(f);
    if (!ok) return ff::fail(FF_ERR_IO, "ff_save_ppm: write to %s failed", path);
    return FF_OK;
}

} // extern "C"

// -------------------------------------------------------------------------------------------------
// Scene description file (the reference's "TODO: Load scene from file", kernel.cu:261).
// -------------------------------------------------------------------------------------------------

struct FfSceneFile {
    std::vector<FfGeometry> geometries;
    std::vector<FfTriangle*> meshes;              // malloc'ed by ff_load_obj
    std::vector<std::pair<std::string, FfBXDF*>> bxdfs; // stable addresses
    FfCamera camera;                               // width/height filled per request
    ~FfSceneFile()
    {
        for (FfTriangle* t : meshes) std::free(t);
        for (auto& b : bxdfs) delete b.second;
    }
};

namespace {

std::vector<std::string> split_ws(const std::string& line)
{
    std::vector<std::string> out;
    size_t i = 0;
    while (i < line.size()) {
        while (i < line.size() && (line[i] == ' ' || line[i] == '\t' || line[i] == '\r' || line[i] == '\n')) ++i;
        if (i >= line.size() || line[i] == '#') break;
        size_t j = i;
        while (j < line.size() && line[j] != ' ' && line[j] != '\t' && line[j] != '\r' && line[j] != '\n') ++j;
        out.push_back(line.substr(i, j - i));
        i = j;
    }
    return out;
}

bool read_floats(const std::vector<std::string>& tok, size_t& i, int n, float* out)
{
    for (int k = 0; k < n; ++k) {
        if (i >= tok.size()) return false;
        char* end = nullptr;
        out[k] = (float)std::strtod(tok[i].c_str(), &end);
        if (end == tok[i].c_str() || *end != '\0') return false;
        ++i;
    }
    return true;
}

} // namespace

extern "C" {

int ff_scene_file_load(const char* path, FfSceneFile** out_scene)
{
    ff::clear_error();
    if (!path || !out_scene) return ff::fail(FF_ERR_INVALID_ARG, "ff_scene_file_load: null argument");
    *out_scene = nullptr;
    FILE* f = std::fopen(path, "rb");
    if (!f) return ff::fail(FF_ERR_IO, "ff_scene_file_load: cannot open '%s': %s", path, std::strerror(errno));
    std::string dir(path);
    const size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);

    FfSceneFile* sc = new FfSceneFile();
    ff_camera_init_default(&sc->camera, 1, 1); // kernel.cu:312-321 literals
    char buf[4096];
    int lineno = 0, status = FF_OK;
    auto bad = [&](const char* what) {
        status = ff::fail(FF_ERR_IO, "%s:%d: %s", path, lineno, what);
    };
    while (status == FF_OK && std::fgets(buf, sizeof buf, f)) {
        ++lineno;
        const std::vector<std::string> tok = split_ws(buf);
        if (tok.empty()) continue;
        size_t i = 1;
        if (tok[0] == "camera") {
            while (status == FF_OK && i < tok.size()) {
                const std::string key = tok[i++];
                float v[3];
                if (key == "position") { if (!read_floats(tok, i, 3, v)) bad("camera position needs 3 numbers"); else sc->camera.m_position = FfVec3{ v[0], v[1], v[2] }; }
                else if (key == "yaw") { if (!read_floats(tok, i, 1, v)) bad("yaw needs a number"); else sc->camera.m_yaw = v[0]; }
                else if (key == "pitch") { if (!read_floats(tok, i, 1, v)) bad("pitch needs a number"); else sc->camera.m_pitch = v[0]; }
                else if (key == "fov") { if (!read_floats(tok, i, 1, v)) bad("fov needs a number"); else sc->camera.m_fov = v[0]; }
                else if (key == "near") { if (!read_floats(tok, i, 1, v)) bad("near needs a number"); else sc->camera.m_nearClip = v[0]; }
                else if (key == "far") { if (!read_floats(tok, i, 1, v)) bad("far needs a number"); else sc->camera.m_farClip = v[0]; }
                else bad("unknown camera key");
            }
        } else if (tok[0] == "bxdf") {
            if (tok.size() < 3) { bad("bxdf needs a name and a type"); break; }
            FfBXDF* b = new FfBXDF();
            ff_bxdf_init(b);
            const std::string& type = tok[2];
            if (type == "diffuse") b->m_type = FF_BXDF_DIFFUSE;
            else if (type == "emitter") b->m_type = FF_BXDF_EMITTER;
            else if (type == "mirror") b->m_type = FF_BXDF_MIRROR;
            else if (type == "glass") b->m_type = FF_BXDF_GLASS;
            else { delete b; bad("unknown bxdf type"); break; }
            sc->bxdfs.emplace_back(tok[1], b);
            i = 3;
            while (status == FF_OK && i < tok.size()) {
                const std::string key = tok[i++];
                float v[3];
                if (key == "albedo") { if (!read_floats(tok, i, 3, v)) bad("albedo needs 3 numbers"); else b->m_albedo = FfVec3{ v[0], v[1], v[2] }; }
                else if (key == "color") { if (!read_floats(tok, i, 3, v)) bad("color needs 3 numbers"); else b->m_emissiveColor = FfVec3{ v[0], v[1], v[2] }; }
                else if (key == "specular") { if (!read_floats(tok, i, 3, v)) bad("specular needs 3 numbers"); else b->m_specularColor = FfVec3{ v[0], v[1], v[2] }; }
                else if (key == "transmittance") { if (!read_floats(tok, i, 3, v)) bad("transmittance needs 3 numbers"); else b->m_transmittanceColor = FfVec3{ v[0], v[1], v[2] }; }
                else if (key == "ior") { if (!read_floats(tok, i, 1, v)) bad("ior needs a number"); else b->m_refractiveIndex = v[0]; }
                else if (key == "intensity") { if (!read_floats(tok, i, 1, v)) bad("intensity needs a number"); else b->m_intensity = v[0]; }
                else bad("unknown bxdf key");
            }
        } else if (tok[0] == "mesh" || tok[0] == "plane" || tok[0] == "sphere") {
            const bool is_mesh = tok[0] == "mesh", is_sphere = tok[0] == "sphere";
            float radius = 0.f;
            FfTriangle* tris = nullptr;
            int ntris = 0;
            if (is_mesh) {
                if (tok.size() < 2) { bad("mesh needs an OBJ path"); break; }
                const std::string obj = (!tok[1].empty() && tok[1][0] == '/') ? tok[1] : dir + tok[1];
                const int st = ff_load_obj(obj.c_str(), &tris, &ntris);
                if (st != FF_OK) { status = st; break; }
                sc->meshes.push_back(tris);
                i = 2;
            } else {
                i = 1;
            }
            FfVec3 pos{ 0, 0, 0 }, rot{ 0, 0, 0 }, scl{ 1, 1, 1 };
            FfBXDF* bx = nullptr;
            while (status == FF_OK && i < tok.size()) {
                const std::string key = tok[i++];
                float v[3];
                if (key == "position") { if (!read_floats(tok, i, 3, v)) bad("position needs 3 numbers"); else pos = FfVec3{ v[0], v[1], v[2] }; }
                else if (key == "rotation") { if (!read_floats(tok, i, 3, v)) bad("rotation needs 3 numbers"); else rot = FfVec3{ v[0], v[1], v[2] }; }
                else if (key == "scale") { if (!read_floats(tok, i, 3, v)) bad("scale needs 3 numbers"); else scl = FfVec3{ v[0], v[1], v[2] }; }
                else if (key == "radius" && is_sphere) { if (!read_floats(tok, i, 1, v) || !(v[0] > 0.f)) bad("radius needs a positive number"); else radius = v[0]; }
                else if (key == "bxdf") {
                    if (i >= tok.size()) { bad("bxdf needs a name"); break; }
                    for (auto& b : sc->bxdfs) if (b.first == tok[i]) bx = b.second;
                    if (!bx) bad("bxdf name not defined above");
                    ++i;
                } else bad("unknown geometry key");
            }
            if (status != FF_OK) break;
            if (!bx) { bad("geometry needs a bxdf"); break; }
            if (is_sphere && !(radius > 0.f)) { bad("sphere needs a radius"); break; }
            FfGeometry g;
            ff_geometry_init(&g, is_mesh ? FF_GEOM_TRIANGLEMESH : (is_sphere ? FF_GEOM_SPHERE : FF_GEOM_PLANE), pos, rot, scl, tris, ntris, radius);
            g.m_bxdf = bx;
            sc->geometries.push_back(g);
        } else {
            bad("unknown statement");
        }
    }
    std::fclose(f);
    if (status == FF_OK && sc->geometries.empty()) status = ff::fail(FF_ERR_IO, "%s: no geometries", path);
    if (status != FF_OK) {
        delete sc;
        return status;
    }
    *out_scene = sc;
    return FF_OK;
}

const FfGeometry* ff_scene_file_geometries(const FfSceneFile* scene, int* out_count)
{
    if (out_count) *out_count = scene ? (int)scene->geometries.size() : 0;
    return scene ? scene->geometries.data() : nullptr;
}

int ff_scene_file_camera(const FfSceneFile* scene, int width, int height, FfCamera* out_camera)
{
    ff::clear_error();
    if (!scene || !out_camera) return ff::fail(FF_ERR_INVALID_ARG, "ff_scene_file_camera: null argument");
    *out_camera = scene->camera;
    out_camera->m_screenWidth = (float)width;
    out_camera->m_screenHeight = (float)height;
    ff_camera_update_basis(out_camera);
    return FF_OK;
}

void ff_scene_file_free(FfSceneFile* scene) { delete scene; }

} // extern "C"
