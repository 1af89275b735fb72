/*
 * oracle/ff_oracle.c — CPU ORACLE. TEST INFRASTRUCTURE ONLY (see ff_oracle.h for the pin status).
 *
 * Plain C99, compiled with -O2 -ffp-contract=off so that every float operation below is a single
 * IEEE-754 binary32 operation in exactly the written order.  Reference citations are relative to
 * /root/reference: K = PathTracer/FireflyEngine/kernel.cu, U = PathTracer/FireflyEngine/utilities.h,
 * GLM = external/include/glm-0.9.9.7.
 */
#include "ff_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------
 * glm restatements
 * ---------------------------------------------------------------------------------------------- */

float orc_radians(float deg) { return deg * 0.01745329251994329576923690768489f; } /* GLM/detail/func_trigonometric.inl:13 */

float orc_dot3(const float* a, const float* b)
{
    /* GLM/detail/func_geometric.inl:52-53: tmp = a*b; tmp.x + tmp.y + tmp.z */
    float tx = a[0] * b[0], ty = a[1] * b[1], tz = a[2] * b[2];
    return (tx + ty) + tz;
}

static float dot4(const float* a, const float* b)
{
    /* GLM/detail/func_geometric.inl:62-63 */
    float tx = a[0] * b[0], ty = a[1] * b[1], tz = a[2] * b[2], tw = a[3] * b[3];
    return (tx + ty) + (tz + tw);
}

void orc_cross3(const float* x, const float* y, float* out)
{
    /* GLM/detail/func_geometric.inl:74-77 */
    float r0 = x[1] * y[2] - y[1] * x[2];
    float r1 = x[2] * y[0] - y[2] * x[0];
    float r2 = x[0] * y[1] - y[0] * x[1];
    out[0] = r0; out[1] = r1; out[2] = r2;
}

static float inversesqrt(float x) { return 1.0f / sqrtf(x); } /* GLM/detail/func_exponential.inl:136-139 */

void orc_normalize3(const float* v, float* out)
{
    /* GLM/detail/func_geometric.inl:88: v * inversesqrt(dot(v, v)) */
    float s = inversesqrt(orc_dot3(v, v));
    out[0] = v[0] * s; out[1] = v[1] * s; out[2] = v[2] * s;
}

static void normalize4(const float* v, float* out)
{
    float s = inversesqrt(dot4(v, v));
    out[0] = v[0] * s; out[1] = v[1] * s; out[2] = v[2] * s; out[3] = v[3] * s;
}

float orc_distance3(const float* p0, const float* p1)
{
    /* GLM/detail/func_geometric.inl:21 length(p1 - p0), :12 sqrt(dot(v, v)) */
    float d[3] = { p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2] };
    return sqrtf(orc_dot3(d, d));
}

void orc_mat4_identity(float* m)
{
    memset(m, 0, 16 * sizeof(float));
    m[0] = m[5] = m[10] = m[15] = 1.0f;
}

void orc_mat4_mul_vec4(const float* m, const float* v, float* out)
{
    /* GLM/detail/type_mat4x4.inl:561-572: (m[0]*v0 + m[1]*v1) + (m[2]*v2 + m[3]*v3) */
    float r[4];
    for (int i = 0; i < 4; ++i) {
        float mul0 = m[0 + i] * v[0];
        float mul1 = m[4 + i] * v[1];
        float add0 = mul0 + mul1;
        float mul2 = m[8 + i] * v[2];
        float mul3 = m[12 + i] * v[3];
        float add1 = mul2 + mul3;
        r[i] = add0 + add1;
    }
    memcpy(out, r, sizeof r);
}

void orc_mat4_mul(const float* a, const float* b, float* out)
{
    /* GLM/detail/type_mat4x4.inl:643-646: ((A0*b0 + A1*b1) + A2*b2) + A3*b3 per column */
    float r[16];
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < 4; ++i) {
            float t = a[0 + i] * b[c * 4 + 0] + a[4 + i] * b[c * 4 + 1];
            t = t + a[8 + i] * b[c * 4 + 2];
            t = t + a[12 + i] * b[c * 4 + 3];
            r[c * 4 + i] = t;
        }
    memcpy(out, r, sizeof r);
}

void orc_mat4_transpose(const float* m, float* out)
{
    float r[16];
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < 4; ++i)
            r[c * 4 + i] = m[i * 4 + c];
    memcpy(out, r, sizeof r);
}

#define M(c, r) m[(c) * 4 + (r)]
void orc_mat4_inverse(const float* m, float* out)
{
    /* GLM/detail/func_matrix.inl:298-349 */
    float Coef00 = M(2,2) * M(3,3) - M(3,2) * M(2,3);
    float Coef02 = M(1,2) * M(3,3) - M(3,2) * M(1,3);
    float Coef03 = M(1,2) * M(2,3) - M(2,2) * M(1,3);

    float Coef04 = M(2,1) * M(3,3) - M(3,1) * M(2,3);
    float Coef06 = M(1,1) * M(3,3) - M(3,1) * M(1,3);
    float Coef07 = M(1,1) * M(2,3) - M(2,1) * M(1,3);

    float Coef08 = M(2,1) * M(3,2) - M(3,1) * M(2,2);
    float Coef10 = M(1,1) * M(3,2) - M(3,1) * M(1,2);
    float Coef11 = M(1,1) * M(2,2) - M(2,1) * M(1,2);

    float Coef12 = M(2,0) * M(3,3) - M(3,0) * M(2,3);
    float Coef14 = M(1,0) * M(3,3) - M(3,0) * M(1,3);
    float Coef15 = M(1,0) * M(2,3) - M(2,0) * M(1,3);

    float Coef16 = M(2,0) * M(3,2) - M(3,0) * M(2,2);
    float Coef18 = M(1,0) * M(3,2) - M(3,0) * M(1,2);
    float Coef19 = M(1,0) * M(2,2) - M(2,0) * M(1,2);

    float Coef20 = M(2,0) * M(3,1) - M(3,0) * M(2,1);
    float Coef22 = M(1,0) * M(3,1) - M(3,0) * M(1,1);
    float Coef23 = M(1,0) * M(2,1) - M(2,0) * M(1,1);

    float Fac0[4] = { Coef00, Coef00, Coef02, Coef03 };
    float Fac1[4] = { Coef04, Coef04, Coef06, Coef07 };
    float Fac2[4] = { Coef08, Coef08, Coef10, Coef11 };
    float Fac3[4] = { Coef12, Coef12, Coef14, Coef15 };
    float Fac4[4] = { Coef16, Coef16, Coef18, Coef19 };
    float Fac5[4] = { Coef20, Coef20, Coef22, Coef23 };

    float Vec0[4] = { M(1,0), M(0,0), M(0,0), M(0,0) };
    float Vec1[4] = { M(1,1), M(0,1), M(0,1), M(0,1) };
    float Vec2[4] = { M(1,2), M(0,2), M(0,2), M(0,2) };
    float Vec3[4] = { M(1,3), M(0,3), M(0,3), M(0,3) };

    static const float SignA[4] = { +1.f, -1.f, +1.f, -1.f };
    static const float SignB[4] = { -1.f, +1.f, -1.f, +1.f };

    float inv[16];
    for (int i = 0; i < 4; ++i) {
        float Inv0 = (Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i]) + Vec3[i] * Fac2[i];
        float Inv1 = (Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i]) + Vec3[i] * Fac4[i];
        float Inv2 = (Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i]) + Vec3[i] * Fac5[i];
        float Inv3 = (Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i]) + Vec2[i] * Fac5[i];
        inv[0 * 4 + i] = Inv0 * SignA[i];
        inv[1 * 4 + i] = Inv1 * SignB[i];
        inv[2 * 4 + i] = Inv2 * SignA[i];
        inv[3 * 4 + i] = Inv3 * SignB[i];
    }
    float Row0[4] = { inv[0], inv[4], inv[8], inv[12] };
    float Dot0[4] = { M(0,0) * Row0[0], M(0,1) * Row0[1], M(0,2) * Row0[2], M(0,3) * Row0[3] };
    float Dot1 = (Dot0[0] + Dot0[1]) + (Dot0[2] + Dot0[3]);
    float OneOverDeterminant = 1.0f / Dot1;
    for (int i = 0; i < 16; ++i) out[i] = inv[i] * OneOverDeterminant;
}
#undef M

void orc_translate(const float* m, const float* v, float* out)
{
    /* GLM/ext/matrix_transform.inl:13-14: Result[3] = m[0]*v[0] + m[1]*v[1] + m[2]*v[2] + m[3] */
    float r[16];
    memcpy(r, m, sizeof r);
    for (int i = 0; i < 4; ++i) {
        float t = m[0 + i] * v[0] + m[4 + i] * v[1];
        t = t + m[8 + i] * v[2];
        r[12 + i] = t + m[12 + i];
    }
    memcpy(out, r, sizeof r);
}

void orc_rotate(const float* m, float angle, const float* v, float* out)
{
    /* GLM/ext/matrix_transform.inl:20-45 */
    float a = angle;
    float c = cosf(a);
    float s = sinf(a);
    float axis[3];
    orc_normalize3(v, axis);
    float temp[3] = { (1.0f - c) * axis[0], (1.0f - c) * axis[1], (1.0f - c) * axis[2] };

    float R00 = c + temp[0] * axis[0];
    float R01 = temp[0] * axis[1] + s * axis[2];
    float R02 = temp[0] * axis[2] - s * axis[1];

    float R10 = temp[1] * axis[0] - s * axis[2];
    float R11 = c + temp[1] * axis[1];
    float R12 = temp[1] * axis[2] + s * axis[0];

    float R20 = temp[2] * axis[0] + s * axis[1];
    float R21 = temp[2] * axis[1] - s * axis[0];
    float R22 = c + temp[2] * axis[2];

    float r[16];
    for (int i = 0; i < 4; ++i) {
        r[0 + i] = (m[0 + i] * R00 + m[4 + i] * R01) + m[8 + i] * R02;
        r[4 + i] = (m[0 + i] * R10 + m[4 + i] * R11) + m[8 + i] * R12;
        r[8 + i] = (m[0 + i] * R20 + m[4 + i] * R21) + m[8 + i] * R22;
        r[12 + i] = m[12 + i];
    }
    memcpy(out, r, sizeof r);
}

void orc_scale(const float* m, const float* v, float* out)
{
    /* GLM/ext/matrix_transform.inl:80-85 */
    float r[16];
    for (int i = 0; i < 4; ++i) {
        r[0 + i] = m[0 + i] * v[0];
        r[4 + i] = m[4 + i] * v[1];
        r[8 + i] = m[8 + i] * v[2];
        r[12 + i] = m[12 + i];
    }
    memcpy(out, r, sizeof r);
}

void orc_look_at_rh(const float* eye, const float* center, const float* up, float* out)
{
    /* GLM/ext/matrix_transform.inl:101-118 */
    float d[3] = { center[0] - eye[0], center[1] - eye[1], center[2] - eye[2] };
    float f[3], s[3], u[3], fxup[3];
    orc_normalize3(d, f);
    orc_cross3(f, up, fxup);
    orc_normalize3(fxup, s);
    orc_cross3(s, f, u);
    orc_mat4_identity(out);
    out[0 * 4 + 0] = s[0];
    out[1 * 4 + 0] = s[1];
    out[2 * 4 + 0] = s[2];
    out[0 * 4 + 1] = u[0];
    out[1 * 4 + 1] = u[1];
    out[2 * 4 + 1] = u[2];
    out[0 * 4 + 2] = -f[0];
    out[1 * 4 + 2] = -f[1];
    out[2 * 4 + 2] = -f[2];
    out[3 * 4 + 0] = -orc_dot3(s, eye);
    out[3 * 4 + 1] = -orc_dot3(u, eye);
    out[3 * 4 + 2] = orc_dot3(f, eye);
}

void orc_perspective_fov_rh_no(float fov, float width, float height, float zNear, float zFar, float* out)
{
    /* GLM/ext/matrix_clip_space.inl:378-388 (GLM_FORCE_DEPTH_ZERO_TO_ONE undefined -> _NO) */
    float rad = fov;
    float h = cosf(0.5f * rad) / sinf(0.5f * rad);
    float w = h * height / width;
    memset(out, 0, 16 * sizeof(float));
    out[0 * 4 + 0] = w;
    out[1 * 4 + 1] = h;
    out[2 * 4 + 2] = -(zFar + zNear) / (zFar - zNear);
    out[2 * 4 + 3] = -1.0f;
    out[3 * 4 + 2] = -(2.0f * zFar * zNear) / (zFar - zNear);
}

/* ------------------------------------------------------------------------------------------------
 * host-side structs
 * ---------------------------------------------------------------------------------------------- */

void orc_geometry_init(FfGeometry* g, int type, FfVec3 pos, FfVec3 rot, FfVec3 scale, FfTriangle* tris, int ntris,
                       float radius)
{
    /* U:176-213 */
    memset(g, 0, sizeof *g);
    g->m_geometryType = type;
    g->m_position = pos;
    g->m_rotation = rot;
    g->m_scale = scale;
    g->m_normal.x = 0.f; g->m_normal.y = 0.f; g->m_normal.z = 1.f; /* U:229 */

    float I[16], translateM[16], rotateM[16], tmp[16], scaleM[16], TR[16];
    static const float X[3] = { 1.f, 0.f, 0.f }, Y[3] = { 0.f, 1.f, 0.f }, Z[3] = { 0.f, 0.f, 1.f };
    orc_mat4_identity(I);
    orc_translate(I, &pos.x, translateM);                       /* U:180 */
    orc_rotate(I, orc_radians(rot.x), X, rotateM);              /* U:182 */
    orc_rotate(I, orc_radians(rot.y), Y, tmp);                  /* U:183 */
    orc_mat4_mul(rotateM, tmp, rotateM);
    orc_rotate(I, orc_radians(rot.z), Z, tmp);                  /* U:184 */
    orc_mat4_mul(rotateM, tmp, rotateM);
    orc_scale(I, &scale.x, scaleM);                             /* U:186 */
    orc_mat4_mul(translateM, rotateM, TR);                      /* U:187 */
    orc_mat4_mul(TR, scaleM, g->m_modelMatrix.m);
    orc_mat4_inverse(g->m_modelMatrix.m, g->m_inverseModelMatrix.m); /* U:189 */

    if (type == FF_GEOM_SPHERE) g->m_sphereRadius = radius;
    if (type == FF_GEOM_TRIANGLEMESH && ntris > 0) {
        g->m_numberOfTriangles = ntris;
        g->m_triangles = tris;
    }
}

void orc_camera_update_basis(FfCamera* c)
{
    /* U:407-418.  cos/sin on float arguments resolve to the float overloads under MSVC's <cmath>. */
    float front[3];
    front[0] = cosf(orc_radians(c->m_yaw)) * cosf(orc_radians(c->m_pitch));
    front[1] = sinf(orc_radians(c->m_pitch));
    front[2] = sinf(orc_radians(c->m_yaw)) * cosf(orc_radians(c->m_pitch));
    float t[3];
    orc_normalize3(front, &c->m_forward.x);
    orc_cross3(&c->m_forward.x, &c->m_worldUp.x, t);
    orc_normalize3(t, &c->m_right.x);
    orc_cross3(&c->m_right.x, &c->m_forward.x, t);
    orc_normalize3(t, &c->m_up.x);
}

void orc_camera_init_default(FfCamera* c, int width, int height)
{
    memset(c, 0, sizeof *c);
    c->m_cameraMovementSpeed = 0.2f;     /* U:287 */
    c->m_cameraMouseSensitivity = 0.2f;  /* U:288 */
    c->m_position.x = 0.f; c->m_position.y = 0.f; c->m_position.z = 15.f;  /* K:312 */
    c->m_forward.x = 0.f; c->m_forward.y = 0.f; c->m_forward.z = -1.f;      /* K:313 */
    c->m_worldUp.x = 0.f; c->m_worldUp.y = 1.f; c->m_worldUp.z = 0.f;       /* K:314 */
    c->m_fov = 70.f;                                                        /* K:315 */
    c->m_screenWidth = (float)width;   /* K:316-317 swap them; un-swapped here (SURVEY hazard 2) */
    c->m_screenHeight = (float)height;
    c->m_nearClip = 0.1f;                                                   /* K:318 */
    c->m_farClip = 1000.f;                                                  /* K:319 */
    c->m_pitch = 0.f;                                                       /* K:320 */
    c->m_yaw = -90.f;                                                       /* K:321 */
    orc_camera_update_basis(c);                                             /* K:322 */
}

void orc_camera_ray_matrix(const FfCamera* c, float* out16)
{
    /* K:203: GetInverseViewMatrix() * GetInverseProjectionMatrix(); U:299-317 */
    float center[3] = { c->m_position.x + c->m_forward.x, c->m_position.y + c->m_forward.y,
                        c->m_position.z + c->m_forward.z };
    float view[16], proj[16], iview[16], iproj[16];
    orc_look_at_rh(&c->m_position.x, center, &c->m_up.x, view);
    orc_mat4_inverse(view, iview);
    orc_perspective_fov_rh_no(orc_radians(c->m_fov), c->m_screenWidth, c->m_screenHeight, c->m_nearClip,
                              c->m_farClip, proj);
    orc_mat4_inverse(proj, iproj);
    orc_mat4_mul(iview, iproj, out16);
}

/* ------------------------------------------------------------------------------------------------
 * device functions
 * ---------------------------------------------------------------------------------------------- */

int orc_intersect_plane(const FfGeometry* plane, const FfRay* ray, FfIntersect* out)
{
    /* K:8-32 */
    const float* n = &plane->m_normal.x;
    const float* o = &ray->m_origin.x;
    const float* d = &ray->m_direction.x;
    float denom = orc_dot3(n, d);                                   /* K:11 */
    if ((double)fabsf(denom) > 1e-7) {                              /* K:12: float vs double literal */
        float p0l0[3] = { -o[0], -o[1], -o[2] };                    /* K:14 */
        float t = orc_dot3(p0l0, n) / denom;                        /* K:15 */
        float P[3] = { o[0] + t * d[0], o[1] + t * d[1], o[2] + t * d[2] }; /* K:16 */
        if (!(P[0] >= -0.5f && P[0] <= 0.5f && P[1] >= -0.5f && P[1] <= 0.5f)) /* K:18 */
            return 0;
        if (t > 0.0f) {                                             /* K:23 */
            out->m_t = t;
            out->m_intersectionPoint.x = P[0]; out->m_intersectionPoint.y = P[1]; out->m_intersectionPoint.z = P[2];
            out->m_normal = plane->m_normal;
            return 1;
        }
        return 0;
    }
    return 0;
}

int orc_intersect_triangle(const FfTriangle* tri, const FfRay* ray, FfIntersect* out)
{
    /* K:35-108 */
    const float EPSILON = 0.000001;                                  /* K:38 (double literal narrowed to float) */
    const float* v0 = &tri->m_v0.x;
    const float* v1 = &tri->m_v1.x;
    const float* v2 = &tri->m_v2.x;
    const float* o = &ray->m_origin.x;
    const float* d = &ray->m_direction.x;
    float edge1[3] = { v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2] }; /* K:44 */
    float edge2[3] = { v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2] }; /* K:45 */
    float Normal[3], pvec[3], qvec[3];
    orc_cross3(edge1, edge2, Normal);                                /* K:48 */
    if (orc_dot3(d, Normal) > 0) return 0;                           /* K:49 */
    orc_cross3(d, edge2, pvec);                                      /* K:53 */
    float det = orc_dot3(edge1, pvec);                               /* K:54 */
    if (det < EPSILON) return 0;                                     /* K:57 */
    float tvec[3] = { o[0] - v0[0], o[1] - v0[1], o[2] - v0[2] };    /* K:61 */
    float u = orc_dot3(tvec, pvec);                                  /* K:62 */
    if (u < 0.0f || u > det) return 0;                               /* K:64 */
    orc_cross3(tvec, edge1, qvec);                                   /* K:68 */
    float v = orc_dot3(d, qvec);                                     /* K:70 */
    if (v < 0.0f || u + v > det) return 0;                           /* K:71 */
    float t = orc_dot3(edge2, qvec);                                 /* K:75 */
    float invDet = (float)(1.0 / (double)det);                       /* K:77: double division narrowed to float */
    t *= invDet;                                                     /* K:79 */
    if (t > EPSILON) {                                               /* K:97 */
        out->m_intersectionPoint.x = o[0] + d[0] * t;                /* K:99 */
        out->m_intersectionPoint.y = o[1] + d[1] * t;
        out->m_intersectionPoint.z = o[2] + d[2] * t;
        out->m_t = t;                                                /* K:100 */
        float nn[3];
        orc_cross3(edge1, edge2, nn);
        orc_normalize3(nn, &out->m_normal.x);                        /* K:101 */
        return 1;
    }
    return 0;
}

int orc_intersect_sphere(const FfGeometry* sphere, const FfRay* ray, FfIntersect* out)
{
    /* SPHERE: the reference declares the type and the radius (U:193-195, U:227) and only printf's at K:166-169.
     * Build-defined: object-space sphere of radius m_sphereRadius about the origin, two-sided (a ray that starts inside
     * hits the far side), same EPSILON as the triangle test, unit normal P / r (reciprocal then multiply, like glm). */
    const float* o = &ray->m_origin.x;
    const float* d = &ray->m_direction.x;
    const float EPSILON = 0.000001; /* the triangle test's constant (K:38) */
    const float rad = sphere->m_sphereRadius;
    const float b = orc_dot3(o, d);
    const float c = orc_dot3(o, o) - rad * rad;
    const float disc = b * b - c;
    if (!(disc >= 0.0f)) return 0;
    const float sq = sqrtf(disc);
    float t = -b - sq;
    if (!(t > EPSILON)) {
        t = -b + sq;
        if (!(t > EPSILON)) return 0;
    }
    out->m_intersectionPoint.x = o[0] + d[0] * t;
    out->m_intersectionPoint.y = o[1] + d[1] * t;
    out->m_intersectionPoint.z = o[2] + d[2] * t;
    out->m_t = t;
    const float inv = 1.0f / rad;
    out->m_normal.x = out->m_intersectionPoint.x * inv;
    out->m_normal.y = out->m_intersectionPoint.y * inv;
    out->m_normal.z = out->m_intersectionPoint.z * inv;
    return 1;
}

int orc_set_intersection(float* tMax, FfIntersect* out, const FfIntersect* obj, const float* model, const FfRay* ray)
{
    /* K:110-125 */
    float P4[4] = { obj->m_intersectionPoint.x, obj->m_intersectionPoint.y, obj->m_intersectionPoint.z, 1.0f };
    float w4[4];
    orc_mat4_mul_vec4(model, P4, w4);                                /* K:113 */
    float distanceOfPOI = orc_distance3(w4, &ray->m_origin.x);       /* K:114 */
    if (distanceOfPOI < *tMax) {                                     /* K:115 */
        float mt[16], nm[16], n4[4] = { obj->m_normal.x, obj->m_normal.y, obj->m_normal.z, 0.f }, r4[4];
        orc_mat4_transpose(model, mt);
        orc_mat4_inverse(mt, nm);
        orc_mat4_mul_vec4(nm, n4, r4);                               /* K:117 */
        out->m_normal.x = r4[0]; out->m_normal.y = r4[1]; out->m_normal.z = r4[2];
        out->m_intersectionPoint.x = w4[0]; out->m_intersectionPoint.y = w4[1]; out->m_intersectionPoint.z = w4[2];
        out->m_t = distanceOfPOI;
        out->m_hit = 1;
        *tMax = distanceOfPOI;
        return 1;
    }
    return 0;
}

static void object_space_ray(const FfGeometry* g, const FfRay* ray, FfRay* out)
{
    /* K:138: Ray(invM * vec4(o,1), normalize(invM * vec4(d,0))) — the normalize acts on the vec4 */
    float o4[4] = { ray->m_origin.x, ray->m_origin.y, ray->m_origin.z, 1.f };
    float d4[4] = { ray->m_direction.x, ray->m_direction.y, ray->m_direction.z, 0.f };
    float ro[4], rd[4], rdn[4];
    orc_mat4_mul_vec4(g->m_inverseModelMatrix.m, o4, ro);
    orc_mat4_mul_vec4(g->m_inverseModelMatrix.m, d4, rd);
    normalize4(rd, rdn);
    out->m_origin.x = ro[0]; out->m_origin.y = ro[1]; out->m_origin.z = ro[2];
    out->m_direction.x = rdn[0]; out->m_direction.y = rdn[1]; out->m_direction.z = rdn[2];
}

static void intersect_rays_counted(const FfRay* ray, const FfGeometry* geoms, int n, FfIntersect* out, OrcCounters* ctr)
{
    /* K:127-176 */
    FfIntersect isect;
    memset(&isect, 0, sizeof isect);
    isect.geometryIndex = -1;                                        /* U:64-65 */
    isect.triangleIndex = -1;
    float tMax = INFINITY;                                           /* K:131 */
    for (int i = 0; i < n; ++i) {                                    /* K:133 */
        const FfGeometry* g = &geoms[i];
        FfRay osr;
        object_space_ray(g, ray, &osr);                              /* K:138 */
        FfIntersect osi;
        memset(&osi, 0, sizeof osi);
        if (g->m_geometryType == FF_GEOM_TRIANGLEMESH) {             /* K:143 */
            for (int j = 0; j < g->m_numberOfTriangles; ++j) {       /* K:145 */
                if (orc_intersect_triangle(&g->m_triangles[j], &osr, &osi)) {
                    if (orc_set_intersection(&tMax, &isect, &osi, g->m_modelMatrix.m, ray)) {
                        isect.geometryIndex = i;                     /* K:151-152 */
                        isect.triangleIndex = j;
                    }
                }
            }
            if (ctr) ctr->tri_tests += (uint64_t)g->m_numberOfTriangles;
        } else if (g->m_geometryType == FF_GEOM_PLANE) {             /* K:157 */
            if (orc_intersect_plane(g, &osr, &osi)) {
                if (orc_set_intersection(&tMax, &isect, &osi, g->m_modelMatrix.m, ray)) {
                    isect.geometryIndex = i;                         /* K:162 */
                    /* deliberate fix: the reference leaves a stale triangleIndex here (unused by it) */
                    isect.triangleIndex = -1;
                }
            }
            if (ctr) ctr->plane_tests += 1;
        } else if (g->m_geometryType == FF_GEOM_SPHERE) {            /* K:166-169 only printf's: build-defined */
            if (orc_intersect_sphere(g, &osr, &osi)) {
                if (orc_set_intersection(&tMax, &isect, &osi, g->m_modelMatrix.m, ray)) {
                    isect.geometryIndex = i;
                    isect.triangleIndex = -1;
                }
            }
            if (ctr) ctr->plane_tests += 1;
        }
    }
    if (ctr) ctr->rays += 1;
    *out = isect;
}

void orc_intersect_rays(const FfRay* ray, const FfGeometry* geoms, int n, FfIntersect* out)
{
    intersect_rays_counted(ray, geoms, n, out, NULL);
}

void orc_primary_ray(const float* cam_mat, const FfCamera* c, int x, int y, FfRay* out)
{
    /* K:197-205 */
    out->m_origin = c->m_position;                                   /* K:198 */
    float Px = ((float)x / c->m_screenWidth) * 2.f - 1.f;            /* K:200 */
    float Py = 1.f - ((float)y / c->m_screenHeight) * 2.f;           /* K:201 */
    float v4[4] = { Px * c->m_farClip, Py * c->m_farClip, 1.f * c->m_farClip, 1.f * c->m_farClip };
    float w4[4];
    orc_mat4_mul_vec4(cam_mat, v4, w4);                              /* K:203 */
    float d[3] = { w4[0] - out->m_origin.x, w4[1] - out->m_origin.y, w4[2] - out->m_origin.z };
    orc_normalize3(d, &out->m_direction.x);                          /* K:205 */
}

/* ------------------------------------------------------------------------------------------------
 * build-defined integrator pieces
 * ---------------------------------------------------------------------------------------------- */

void orc_philox2x32_10(uint32_t c0, uint32_t c1, uint32_t key, uint32_t* out0, uint32_t* out1)
{
    /* Philox2x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11):
     * multiplier 0xD256D193, Weyl key increment 0x9E3779B9. */
    for (int r = 0; r < 10; ++r) {
        if (r > 0) key += 0x9E3779B9u;
        uint64_t p = (uint64_t)0xD256D193u * (uint64_t)c0;
        uint32_t hi = (uint32_t)(p >> 32), lo = (uint32_t)p;
        c0 = hi ^ key ^ c1;
        c1 = lo;
    }
    *out0 = c0;
    *out1 = c1;
}

void orc_sample_uniforms(uint32_t pixel_index, uint32_t sample, uint32_t bounce, uint64_t seed, float* u1, float* u2,
                         uint32_t* raw1)
{
    /* counter = (global pixel index, sample<<8 | bounce), key = seed folded to 32 bits */
    uint32_t key = (uint32_t)seed ^ (uint32_t)(seed >> 32);
    uint32_t r0, r1;
    orc_philox2x32_10(pixel_index, (sample << 8) | (bounce & 0xFFu), key, &r0, &r1);
    *u1 = (float)(r0 >> 8) * 5.9604644775390625e-08f; /* 2^-24, exact */
    *u2 = (float)(r1 >> 8) * 5.9604644775390625e-08f;
    if (raw1) *raw1 = r1 >> 8;
}

void orc_cosine_sample_hemisphere(float u1, uint32_t k24, float* out)
{
    /* U:46-55: r = sqrt(u1); theta = 2*pi*u2; (r cos theta, r sin theta, sqrt(max(0, 1-u1))).
     * u2 = k24 * 2^-24.  sin/cos of theta are evaluated with an exact octant reduction on the integer k24 and
     * fixed-order float polynomials on [0, pi/4], so the CPU and GPU results are bit-identical. */
    uint32_t oct = k24 >> 21;
    uint32_t f = k24 & 0x1FFFFFu;
    uint32_t mfrac = (oct & 1u) ? (0x200000u - f) : f;
    float a = (float)mfrac * 3.7450704e-07f;         /* (pi/4) * 2^-21 rounded to float = 0x1.921fb6p-22 */
    float a2 = a * a;
    /* sin(a) = a + a*a2*(S1 + a2*(S2 + a2*(S3 + a2*S4))) */
    float sp = -1.9841270e-04f + a2 * 2.7557319e-06f;
    sp = 8.3333333e-03f + a2 * sp;
    sp = -1.6666667e-01f + a2 * sp;
    float s = a + (a * a2) * sp;
    /* cos(a) = 1 + a2*(C1 + a2*(C2 + a2*(C3 + a2*C4))) */
    float cp = -1.3888889e-03f + a2 * 2.4801587e-05f;
    cp = 4.1666667e-02f + a2 * cp;
    cp = -0.5f + a2 * cp;
    float c = 1.0f + a2 * cp;
    float sn, cs;
    if ((oct + 1u) & 2u) { sn = c; cs = s; } else { sn = s; cs = c; }
    if (oct >= 4u) sn = -sn;
    if (oct >= 2u && oct <= 5u) cs = -cs;
    float r = sqrtf(u1);
    out[0] = r * cs;
    out[1] = r * sn;
    out[2] = sqrtf(fmaxf(0.0f, 1.0f - u1));
}

void orc_onb(const float* n, float* t, float* b)
{
    /* Duff et al., "Building an Orthonormal Basis, Revisited", JCGT 2017 */
    float sign = copysignf(1.0f, n[2]);
    float a = -1.0f / (sign + n[2]);
    float bb = (n[0] * n[1]) * a;
    t[0] = 1.0f + ((sign * n[0]) * n[0]) * a;
    t[1] = sign * bb;
    t[2] = -sign * n[0];
    b[0] = bb;
    b[1] = sign + (n[1] * n[1]) * a;
    b[2] = -n[1];
}

#define ORC_RAY_EPS 1.0e-4f

static uint8_t to_u8(float v)
{
    /* K:214 float -> unsigned char truncation; values >= 256 are UB in the reference, clamped here */
    float s = v * 255.f;
    if (!(s > 0.f)) return 0;
    if (s >= 255.f) return 255;
    return (uint8_t)s;
}

static void shade_pixel(const FfGeometry* geoms, int n, const FfCamera* cam, const float* cam_mat,
                        const FfRenderParams* p, int x, int y, float* rad3, int* any_hit, OrcCounters* ctr)
{
    FfRay primary;
    orc_primary_ray(cam_mat, cam, x, y, &primary);
    uint32_t pixel_index = (uint32_t)(y * p->width + x);            /* K:191 */
    *any_hit = 0;

    if (p->shade_mode == FF_SHADE_NORMAL_DEBUG) {
        /* K:207-214 + shade() K:178-184 */
        FfIntersect is;
        intersect_rays_counted(&primary, geoms, n, &is, ctr);
        if (is.m_hit) {
            rad3[0] = fabsf(is.m_normal.x);
            rad3[1] = fabsf(is.m_normal.y);
            rad3[2] = fabsf(is.m_normal.z);
            *any_hit = 1;
        } else {
            rad3[0] = rad3[1] = rad3[2] = 0.f;
        }
        return;
    }

    /* Samples are accumulated in blocks: a block sums its samples sequentially from 0, the blocks of a pixel are added
     * in order (block size: a multiple of 64, 64 up to 1024 spp).  The order is part of the result's definition. */
    const int block_spp = 64 * ((p->spp + 1023) / 1024);
    float total[3] = { 0.f, 0.f, 0.f };
    float acc[3] = { 0.f, 0.f, 0.f };
    for (int s = 0; s < p->spp; ++s) {
        if (s > 0 && s % block_spp == 0) {
            total[0] = total[0] + acc[0]; total[1] = total[1] + acc[1]; total[2] = total[2] + acc[2];
            acc[0] = acc[1] = acc[2] = 0.f;
        }
        FfRay ray = primary;
        float beta[3] = { 1.f, 1.f, 1.f };
        float L[3] = { 0.f, 0.f, 0.f };
        for (int b = 0; b < p->bounces; ++b) {
            FfIntersect is;
            intersect_rays_counted(&ray, geoms, n, &is, ctr);
            if (!is.m_hit) break;
            const FfBXDF* bx = geoms[is.geometryIndex].m_bxdf;
            if (bx->m_type == FF_BXDF_EMITTER) {
                /* U:96-103: m_emissiveColor * m_intensity, two-sided */
                float Le[3] = { bx->m_emissiveColor.x * bx->m_intensity, bx->m_emissiveColor.y * bx->m_intensity,
                                bx->m_emissiveColor.z * bx->m_intensity };
                L[0] = L[0] + beta[0] * Le[0];
                L[1] = L[1] + beta[1] * Le[1];
                L[2] = L[2] + beta[2] * Le[2];
                break;
            }
            /* MIRROR (U:68-75 declares it, U:108 leaves it as a TODO; build-defined): perfect specular reflection about the
             * shading normal, throughput *= m_specularColor, no random numbers consumed.
             * GLASS (same status): smooth dielectric of index m_refractiveIndex; the facing side decides entering/leaving,
             * unpolarised Fresnel reflectance F, one uniform of the bounce's random pair picks reflection (probability F,
             * throughput *= m_specularColor) or refraction (throughput *= m_transmittanceColor); total internal
             * reflection reflects.  Everything else is diffuse (U:109's constant-true test). */
            const int mirror = bx->m_type == FF_BXDF_MIRROR;
            const int glass = bx->m_type == FF_BXDF_GLASS;
            if (!glass) {
                const FfVec3 tint = mirror ? bx->m_specularColor : bx->m_albedo;
                beta[0] = beta[0] * tint.x;
                beta[1] = beta[1] * tint.y;
                beta[2] = beta[2] * tint.z;
            }
            if (b == p->bounces - 1) break;
            /* Shading normal: the direction of inverse(transpose(M)) * n_obj, normalised once.  For triangles the
             * un-normalised face normal cross(e1, e2) is transformed (the reference's Intersect carries its normalised
             * value, K:101, which NORMAL_DEBUG shades; the direction is the same and one sqrt/divide per bounce is saved). */
            float nobj[3], nw4[4], nrm[3];
            const FfGeometry* hg = &geoms[is.geometryIndex];
            if (is.triangleIndex >= 0) {
                const FfTriangle* tr = &hg->m_triangles[is.triangleIndex];
                float e1[3] = { tr->m_v1.x - tr->m_v0.x, tr->m_v1.y - tr->m_v0.y, tr->m_v1.z - tr->m_v0.z };
                float e2[3] = { tr->m_v2.x - tr->m_v0.x, tr->m_v2.y - tr->m_v0.y, tr->m_v2.z - tr->m_v0.z };
                orc_cross3(e1, e2, nobj);
                if (p->shade_mode == FF_SHADE_DIFFUSE_PATH_SMOOTH) {
                    /* Interpolated vertex normals (U:163-170) with the barycentrics the reference computes and drops
                     * (K:62, K:70, K:80-81): n = ((1 - u) - v) n0 + u n1 + v n2.  All-zero vertex normals (an OBJ
                     * without vn) keep the geometric normal. */
                    FfRay osr;
                    object_space_ray(hg, &ray, &osr);
                    const float* o = &osr.m_origin.x;
                    const float* d = &osr.m_direction.x;
                    float pvec[3], qvec[3];
                    orc_cross3(d, e2, pvec);
                    const float det = orc_dot3(e1, pvec);
                    const float tvec[3] = { o[0] - tr->m_v0.x, o[1] - tr->m_v0.y, o[2] - tr->m_v0.z };
                    float u = orc_dot3(tvec, pvec);
                    orc_cross3(tvec, e1, qvec);
                    float v = orc_dot3(d, qvec);
                    const float invDet = (float)(1.0 / (double)det);
                    u = u * invDet;
                    v = v * invDet;
                    const float w = (1.0f - u) - v;
                    const float sn[3] = { (w * tr->m_n0.x + u * tr->m_n1.x) + v * tr->m_n2.x,
                                          (w * tr->m_n0.y + u * tr->m_n1.y) + v * tr->m_n2.y,
                                          (w * tr->m_n0.z + u * tr->m_n1.z) + v * tr->m_n2.z };
                    if (!(sn[0] == 0.0f && sn[1] == 0.0f && sn[2] == 0.0f)) { nobj[0] = sn[0]; nobj[1] = sn[1]; nobj[2] = sn[2]; }
                }
            } else if (hg->m_geometryType == FF_GEOM_SPHERE) {
                /* the object-space unit normal of the hit: intersect the winning sphere again (same arithmetic, same result) */
                FfRay osr;
                FfIntersect osi;
                object_space_ray(hg, &ray, &osr);
                memset(&osi, 0, sizeof osi);
                orc_intersect_sphere(hg, &osr, &osi);
                nobj[0] = osi.m_normal.x; nobj[1] = osi.m_normal.y; nobj[2] = osi.m_normal.z;
            } else {
                nobj[0] = hg->m_normal.x; nobj[1] = hg->m_normal.y; nobj[2] = hg->m_normal.z;
            }
            {
                float mt[16], nm[16], n4[4] = { nobj[0], nobj[1], nobj[2], 0.f };
                orc_mat4_transpose(hg->m_modelMatrix.m, mt);
                orc_mat4_inverse(mt, nm);
                orc_mat4_mul_vec4(nm, n4, nw4);
            }
            orc_normalize3(nw4, nrm);
            int flipped = 0;
            if (orc_dot3(nrm, &ray.m_direction.x) > 0.f) { nrm[0] = -nrm[0]; nrm[1] = -nrm[1]; nrm[2] = -nrm[2]; flipped = 1; }
            float wo[3];
            float side[3] = { nrm[0], nrm[1], nrm[2] }; /* the next ray starts on this side of the surface */
            if (glass) {
                const float* d = &ray.m_direction.x;
                const float ior = bx->m_refractiveIndex;
                const float eta = flipped ? ior : 1.0f / ior; /* n_incident / n_transmitted: the geometric normal faces the outside */
                const float ci = -orc_dot3(nrm, d);
                const float s2 = (eta * eta) * (1.0f - ci * ci);
                int reflect = 1;
                float ct = 0.f;
                if (s2 < 1.0f) {
                    ct = sqrtf(1.0f - s2);
                    const float a = eta * ci, bq = eta * ct;
                    const float rs = (a - ct) / (a + ct), rp = (ci - bq) / (ci + bq);
                    const float F = 0.5f * (rs * rs + rp * rp);
                    float u1, u2;
                    uint32_t k24;
                    orc_sample_uniforms(pixel_index, (uint32_t)s, (uint32_t)b, p->seed, &u1, &u2, &k24);
                    (void)u2; (void)k24;
                    reflect = u1 < F;
                }
                FfVec3 tint;
                if (reflect) {
                    const float k2 = 2.0f * ci;
                    wo[0] = d[0] + k2 * nrm[0];
                    wo[1] = d[1] + k2 * nrm[1];
                    wo[2] = d[2] + k2 * nrm[2];
                    tint = bx->m_specularColor;
                } else {
                    const float k = eta * ci - ct;
                    wo[0] = eta * d[0] + k * nrm[0];
                    wo[1] = eta * d[1] + k * nrm[1];
                    wo[2] = eta * d[2] + k * nrm[2];
                    tint = bx->m_transmittanceColor;
                    side[0] = -nrm[0]; side[1] = -nrm[1]; side[2] = -nrm[2];
                }
                beta[0] = beta[0] * tint.x;
                beta[1] = beta[1] * tint.y;
                beta[2] = beta[2] * tint.z;
            } else if (mirror) {
                /* wo = d - (2 (n.d)) n with the facing normal: the reflected direction keeps the length of d */
                const float k2 = 2.0f * orc_dot3(nrm, &ray.m_direction.x);
                wo[0] = ray.m_direction.x - k2 * nrm[0];
                wo[1] = ray.m_direction.y - k2 * nrm[1];
                wo[2] = ray.m_direction.z - k2 * nrm[2];
            } else {
                float u1, u2, wl[3], tt[3], bb[3];
                uint32_t k24;
                orc_sample_uniforms(pixel_index, (uint32_t)s, (uint32_t)b, p->seed, &u1, &u2, &k24);
                (void)u2;
                orc_cosine_sample_hemisphere(u1, k24, wl);
                orc_onb(nrm, tt, bb);
                /* the local direction is unit and the basis orthonormal: the world direction is used as is (|wo| = 1 +- 1e-6) */
                wo[0] = (tt[0] * wl[0] + bb[0] * wl[1]) + nrm[0] * wl[2];
                wo[1] = (tt[1] * wl[0] + bb[1] * wl[1]) + nrm[1] * wl[2];
                wo[2] = (tt[2] * wl[0] + bb[2] * wl[1]) + nrm[2] * wl[2];
            }
            ray.m_origin.x = is.m_intersectionPoint.x + side[0] * ORC_RAY_EPS;
            ray.m_origin.y = is.m_intersectionPoint.y + side[1] * ORC_RAY_EPS;
            ray.m_origin.z = is.m_intersectionPoint.z + side[2] * ORC_RAY_EPS;
            ray.m_direction.x = wo[0]; ray.m_direction.y = wo[1]; ray.m_direction.z = wo[2];
        }
        acc[0] = acc[0] + L[0];
        acc[1] = acc[1] + L[1];
        acc[2] = acc[2] + L[2];
    }
    total[0] = total[0] + acc[0]; total[1] = total[1] + acc[1]; total[2] = total[2] + acc[2];
    float inv = 1.0f / (float)p->spp;
    rad3[0] = total[0] * inv;
    rad3[1] = total[1] * inv;
    rad3[2] = total[2] * inv;
    *any_hit = 1;
}

void orc_render(const FfGeometry* geoms, int n, const FfCamera* cam, const FfRenderParams* params, int x0, int y0,
                int w, int h, uint8_t* rgb8, float* radiance, OrcCounters* counters, int threads)
{
    float cam_mat[16];
    orc_camera_ray_matrix(cam, cam_mat);
    int xlim = params->width, ylim = params->height;
    if (params->grid_mode == FF_GRID_REFERENCE_FLOOR) {               /* K:306-309 */
        xlim = (params->width / 16) * 16;
        ylim = (params->height / 16) * 16;
    }
    OrcCounters total = { 0, 0, 0 };
    if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
    {
        OrcCounters local = { 0, 0, 0 };
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int ly = 0; ly < h; ++ly) {
            int y = y0 + ly;
            for (int lx = 0; lx < w; ++lx) {
                int x = x0 + lx;
                float rad[3] = { 0.f, 0.f, 0.f };
                int hit = 0;
                if (x < xlim && y < ylim) shade_pixel(geoms, n, cam, cam_mat, params, x, y, rad, &hit, &local);
                size_t o = ((size_t)ly * (size_t)w + (size_t)lx) * 3;
                if (radiance) { radiance[o] = rad[0]; radiance[o + 1] = rad[1]; radiance[o + 2] = rad[2]; }
                if (rgb8) { rgb8[o] = to_u8(rad[0]); rgb8[o + 1] = to_u8(rad[1]); rgb8[o + 2] = to_u8(rad[2]); }
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            total.rays += local.rays;
            total.tri_tests += local.tri_tests;
            total.plane_tests += local.plane_tests;
        }
    }
    if (counters) *counters = total;
}
