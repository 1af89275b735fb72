// ff_math.h — host-side vector/matrix arithmetic with glm 0.9.9.7's exact operation order.
//
// The reference builds its model/camera matrices with glm on the host (utilities.h:180-189, 299-317,
// 407-418) and multiplies by them on the device; float results depend on the association order glm uses.
// Everything here follows that order (citations: GLM = external/include/glm-0.9.9.7 in the reference) and the
// library is compiled with -ffp-contract=off, so matrices built here are bit-identical to glm's.
#pragma once

#include <cmath>
#include <cstring>

namespace ffm {

struct V3 {
    float x, y, z;
};
struct V4 {
    float x, y, z, w;
};
// column-major 4x4: c[col] is a column vector (glm::mat4::operator[])
struct M4 {
    V4 c[4];
};

inline V3 v3(float x, float y, float z) { return V3{ x, y, z }; }
inline V4 v4(float x, float y, float z, float w) { return V4{ x, y, z, w }; }
inline V4 v4(const V3& a, float w) { return V4{ a.x, a.y, a.z, w }; }
inline V3 xyz(const V4& a) { return V3{ a.x, a.y, a.z }; }

inline V3 operator+(const V3& a, const V3& b) { return V3{ a.x + b.x, a.y + b.y, a.z + b.z }; }
inline V3 operator-(const V3& a, const V3& b) { return V3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
inline V3 operator*(const V3& a, float s) { return V3{ a.x * s, a.y * s, a.z * s }; }
inline V3 operator-(const V3& a) { return V3{ -a.x, -a.y, -a.z }; }
inline V4 operator+(const V4& a, const V4& b) { return V4{ a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w }; }
inline V4 operator-(const V4& a, const V4& b) { return V4{ a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w }; }
inline V4 operator*(const V4& a, float s) { return V4{ a.x * s, a.y * s, a.z * s, a.w * s }; }
inline V4 operator*(const V4& a, const V4& b) { return V4{ a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w }; }

// GLM/detail/func_geometric.inl:48-55
inline float dot(const V3& a, const V3& b)
{
    const float px = a.x * b.x, py = a.y * b.y, pz = a.z * b.z;
    return (px + py) + pz;
}
// GLM/detail/func_geometric.inl:68-79
inline V3 cross(const V3& a, const V3& b)
{
    return V3{ a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y };
}
// GLM/detail/func_geometric.inl:82-90 with inversesqrt = 1/sqrt (func_exponential.inl:136-139)
inline V3 normalize(const V3& a)
{
    const float inv = 1.0f / std::sqrt(dot(a, a));
    return a * inv;
}
// GLM/detail/func_trigonometric.inl:13
inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

inline M4 identity()
{
    M4 r;
    r.c[0] = v4(1, 0, 0, 0);
    r.c[1] = v4(0, 1, 0, 0);
    r.c[2] = v4(0, 0, 1, 0);
    r.c[3] = v4(0, 0, 0, 1);
    return r;
}

// GLM/detail/type_mat4x4.inl:561-572 — (m0*v0 + m1*v1) + (m2*v2 + m3*v3)
inline V4 mul(const M4& m, const V4& v)
{
    const V4 lo = m.c[0] * v.x + m.c[1] * v.y;
    const V4 hi = m.c[2] * v.z + m.c[3] * v.w;
    return lo + hi;
}

// GLM/detail/type_mat4x4.inl:630-648 — ((A0*b0 + A1*b1) + A2*b2) + A3*b3
inline M4 mul(const M4& a, const M4& b)
{
    M4 r;
    for (int k = 0; k < 4; ++k) {
        const V4& bc = b.c[k];
        r.c[k] = ((a.c[0] * bc.x + a.c[1] * bc.y) + a.c[2] * bc.z) + a.c[3] * bc.w;
    }
    return r;
}

inline float at(const M4& m, int col, int row) { return (&m.c[col].x)[row]; }

// GLM/detail/func_matrix.inl:170-196
inline M4 transpose(const M4& m)
{
    M4 r;
    for (int k = 0; k < 4; ++k) r.c[k] = v4(at(m, 0, k), at(m, 1, k), at(m, 2, k), at(m, 3, k));
    return r;
}

// GLM/detail/func_matrix.inl:294-350 (cofactor expansion, scaled by 1/det with det = dot(m[0], row0))
inline M4 inverse(const M4& m)
{
    auto e = [&](int c, int r) { return at(m, c, r); };
    const float s00 = e(2, 2) * e(3, 3) - e(3, 2) * e(2, 3);
    const float s02 = e(1, 2) * e(3, 3) - e(3, 2) * e(1, 3);
    const float s03 = e(1, 2) * e(2, 3) - e(2, 2) * e(1, 3);
    const float s04 = e(2, 1) * e(3, 3) - e(3, 1) * e(2, 3);
    const float s06 = e(1, 1) * e(3, 3) - e(3, 1) * e(1, 3);
    const float s07 = e(1, 1) * e(2, 3) - e(2, 1) * e(1, 3);
    const float s08 = e(2, 1) * e(3, 2) - e(3, 1) * e(2, 2);
    const float s10 = e(1, 1) * e(3, 2) - e(3, 1) * e(1, 2);
    const float s11 = e(1, 1) * e(2, 2) - e(2, 1) * e(1, 2);
    const float s12 = e(2, 0) * e(3, 3) - e(3, 0) * e(2, 3);
    const float s14 = e(1, 0) * e(3, 3) - e(3, 0) * e(1, 3);
    const float s15 = e(1, 0) * e(2, 3) - e(2, 0) * e(1, 3);
    const float s16 = e(2, 0) * e(3, 2) - e(3, 0) * e(2, 2);
    const float s18 = e(1, 0) * e(3, 2) - e(3, 0) * e(1, 2);
    const float s19 = e(1, 0) * e(2, 2) - e(2, 0) * e(1, 2);
    const float s20 = e(2, 0) * e(3, 1) - e(3, 0) * e(2, 1);
    const float s22 = e(1, 0) * e(3, 1) - e(3, 0) * e(1, 1);
    const float s23 = e(1, 0) * e(2, 1) - e(2, 0) * e(1, 1);

    const V4 f0 = v4(s00, s00, s02, s03), f1 = v4(s04, s04, s06, s07), f2 = v4(s08, s08, s10, s11);
    const V4 f3 = v4(s12, s12, s14, s15), f4 = v4(s16, s16, s18, s19), f5 = v4(s20, s20, s22, s23);
    const V4 a0 = v4(e(1, 0), e(0, 0), e(0, 0), e(0, 0));
    const V4 a1 = v4(e(1, 1), e(0, 1), e(0, 1), e(0, 1));
    const V4 a2 = v4(e(1, 2), e(0, 2), e(0, 2), e(0, 2));
    const V4 a3 = v4(e(1, 3), e(0, 3), e(0, 3), e(0, 3));

    const V4 i0 = (a1 * f0 - a2 * f1) + a3 * f2;
    const V4 i1 = (a0 * f0 - a2 * f3) + a3 * f4;
    const V4 i2 = (a0 * f1 - a1 * f3) + a3 * f5;
    const V4 i3 = (a0 * f2 - a1 * f4) + a2 * f5;
    const V4 sa = v4(+1, -1, +1, -1), sb = v4(-1, +1, -1, +1);
    M4 inv;
    inv.c[0] = i0 * sa;
    inv.c[1] = i1 * sb;
    inv.c[2] = i2 * sa;
    inv.c[3] = i3 * sb;
    const V4 row0 = v4(inv.c[0].x, inv.c[1].x, inv.c[2].x, inv.c[3].x);
    const V4 d = m.c[0] * row0;
    const float det = (d.x + d.y) + (d.z + d.w);
    const float rdet = 1.0f / det;
    for (int k = 0; k < 4; ++k) inv.c[k] = inv.c[k] * rdet;
    return inv;
}

// GLM/ext/matrix_transform.inl:10-16
inline M4 translate(const M4& m, const V3& v)
{
    M4 r = m;
    r.c[3] = ((m.c[0] * v.x + m.c[1] * v.y) + m.c[2] * v.z) + m.c[3];
    return r;
}

// GLM/ext/matrix_transform.inl:18-46
inline M4 rotate(const M4& m, float angle, const V3& v)
{
    const float c = std::cos(angle), s = std::sin(angle);
    const V3 axis = normalize(v);
    const V3 t = axis * (1.0f - c); // glm: (T(1) - c) * axis — scalar*vec is commutative per component
    const float r00 = c + t.x * axis.x, r01 = t.x * axis.y + s * axis.z, r02 = t.x * axis.z - s * axis.y;
    const float r10 = t.y * axis.x - s * axis.z, r11 = c + t.y * axis.y, r12 = t.y * axis.z + s * axis.x;
    const float r20 = t.z * axis.x + s * axis.y, r21 = t.z * axis.y - s * axis.x, r22 = c + t.z * axis.z;
    M4 r;
    r.c[0] = (m.c[0] * r00 + m.c[1] * r01) + m.c[2] * r02;
    r.c[1] = (m.c[0] * r10 + m.c[1] * r11) + m.c[2] * r12;
    r.c[2] = (m.c[0] * r20 + m.c[1] * r21) + m.c[2] * r22;
    r.c[3] = m.c[3];
    return r;
}

// GLM/ext/matrix_transform.inl:77-86
inline M4 scale(const M4& m, const V3& v)
{
    M4 r;
    r.c[0] = m.c[0] * v.x;
    r.c[1] = m.c[1] * v.y;
    r.c[2] = m.c[2] * v.z;
    r.c[3] = m.c[3];
    return r;
}

// GLM/ext/matrix_transform.inl:99-119
inline M4 lookAtRH(const V3& eye, const V3& center, const V3& up)
{
    const V3 f = normalize(center - eye);
    const V3 s = normalize(cross(f, up));
    const V3 u = cross(s, f);
    M4 r = identity();
    r.c[0].x = s.x; r.c[1].x = s.y; r.c[2].x = s.z;
    r.c[0].y = u.x; r.c[1].y = u.y; r.c[2].y = u.z;
    r.c[0].z = -f.x; r.c[1].z = -f.y; r.c[2].z = -f.z;
    r.c[3].x = -dot(s, eye);
    r.c[3].y = -dot(u, eye);
    r.c[3].z = dot(f, eye);
    return r;
}

// GLM/ext/matrix_clip_space.inl:372-389 (depth -1..1: GLM_FORCE_DEPTH_ZERO_TO_ONE is not defined by the reference)
inline M4 perspectiveFovRH_NO(float fov, float width, float height, float zNear, float zFar)
{
    const float h = std::cos(0.5f * fov) / std::sin(0.5f * fov);
    const float w = h * height / width;
    M4 r;
    std::memset(&r, 0, sizeof r);
    r.c[0].x = w;
    r.c[1].y = h;
    r.c[2].z = -(zFar + zNear) / (zFar - zNear);
    r.c[2].w = -1.0f;
    r.c[3].z = -(2.0f * zFar * zNear) / (zFar - zNear);
    return r;
}

inline M4 load(const float* p)
{
    M4 r;
    std::memcpy(&r, p, sizeof r);
    return r;
}
inline void store(const M4& m, float* p) { std::memcpy(p, &m, sizeof m); }

} // namespace ffm
