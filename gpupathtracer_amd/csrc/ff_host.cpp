// ff_host.cpp — host-side pieces of the ABI that need no GPU: error strings, the reference's struct
// constructors (Geometry, Camera, BXDF) restated with glm's operation order, and the OBJ reader.
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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

} // extern "C"
