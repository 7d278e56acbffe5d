// ORACLE -- test infrastructure only.  Nothing under oracle/ is linked, imported or
// executed by the product (rgk_amd/); only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may use it, and only as the checker.
//
// Minimal float vector/quaternion/matrix algebra with the semantics of the GLM
// (<= 0.9.8, unvendored, unpinned) calls the reference makes on the hot path.
// GLM is absent from /root/reference and from this image, so these formulas are a
// restatement of GLM's published definitions; parity at this boundary is UNPINNED
// (SURVEY 8c "Third-party arithmetic on the path").  Call sites followed:
//   src/glm.hpp:18-35, src/glm.cpp:3-59, src/LTC/ltc.cpp:59-143, src/texture.cpp:38-39.
#pragma once
#include <cmath>
#include <cstdint>

#include "../include/rgk_libm.h" // sin / cos / acos / asin / atan2 on the path: pinned definitions shared with the HIP kernels

namespace orc {

struct vec2 {
    float x, y;
    vec2() : x(0), y(0) {}
    vec2(float x_, float y_) : x(x_), y(y_) {}
    float operator[](int i) const { return i == 0 ? x : y; }
};
inline vec2 operator*(vec2 a, float s) { return vec2(a.x * s, a.y * s); }
inline vec2 operator*(float s, vec2 a) { return vec2(a.x * s, a.y * s); }
inline vec2 operator+(vec2 a, vec2 b) { return vec2(a.x + b.x, a.y + b.y); }

struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, vec3 a) { return vec3(s * a.x, s * a.y, s * a.z); }
inline vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator/(vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }

// glm::dot<vec3>: tmp = a*b; tmp.x + tmp.y + tmp.z
inline float dot(vec3 a, vec3 b) {
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
// glm::cross
inline vec3 cross(vec3 x, vec3 y) {
    return vec3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
inline float length(vec3 v) { return std::sqrt(dot(v, v)); }
inline float length(vec2 v) { return std::sqrt(v.x * v.x + v.y * v.y); }
// glm::normalize: v * inversesqrt(dot(v,v)), inversesqrt(x) = 1/sqrt(x)
inline vec3 normalize(vec3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
inline float distance2(vec3 a, vec3 b) { vec3 d = b - a; return dot(d, d); }
inline float clampf(float x, float lo, float hi) { return std::fmin(std::fmax(x, lo), hi); }
// glm::angle(x,y) = acos(clamp(dot(x,y), -1, 1))
inline float angle(vec3 a, vec3 b) { return rgk_acosf(clampf(dot(a, b), -1.0f, 1.0f)); } // acos pinned: include/rgk_libm.h
inline vec3 vabs(vec3 v) { return vec3(std::fabs(v.x), std::fabs(v.y), std::fabs(v.z)); }
// glm::repeat(x) = fract(x) = x - floor(x)
inline float repeat(float x) { return x - std::floor(x); }

constexpr float PI_F = 3.14159265358979323846264338327950288f; // glm::pi<float>()

struct quat {
    float w, x, y, z;
    quat() : w(1), x(0), y(0), z(0) {}
    quat(float w_, float x_, float y_, float z_) : w(w_), x(x_), y(y_), z(z_) {}
};
// glm: operator*(quat, vec3): uv = cross(qv, v); uuv = cross(qv, uv); v + ((uv*w) + uuv)*2
inline vec3 operator*(const quat& q, vec3 v) {
    vec3 qv(q.x, q.y, q.z);
    vec3 uv = cross(qv, v);
    vec3 uuv = cross(qv, uv);
    return v + ((uv * q.w) + uuv) * 2.0f;
}
// glm::inverse(quat) = conjugate(q) / dot(q,q)
inline quat inverse(const quat& q) {
    float d = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w; // glm dot(vec4)-style order differs by ulps: unpinned
    return quat(q.w / d, -q.x / d, -q.y / d, -q.z / d);
}
// glm::angleAxis(angle, axis)
inline quat angleAxis(float a, vec3 axis) {
    float s = rgk_sinf(a * 0.5f);
    return quat(rgk_cosf(a * 0.5f), axis.x * s, axis.y * s, axis.z * s);
}

// column-major 3x3, m[col][row] like glm::mat3
struct mat3 {
    float m[3][3];
    mat3() { for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) m[c][r] = (c == r) ? 1.0f : 0.0f; }
    mat3(vec3 c0, vec3 c1, vec3 c2) {
        m[0][0] = c0.x; m[0][1] = c0.y; m[0][2] = c0.z;
        m[1][0] = c1.x; m[1][1] = c1.y; m[1][2] = c1.z;
        m[2][0] = c2.x; m[2][1] = c2.y; m[2][2] = c2.z;
    }
};
// glm mat3 * vec3: m[0]*v.x + m[1]*v.y + m[2]*v.z, component-wise left to right
inline vec3 operator*(const mat3& a, vec3 v) {
    return vec3(a.m[0][0] * v.x + a.m[1][0] * v.y + a.m[2][0] * v.z,
                a.m[0][1] * v.x + a.m[1][1] * v.y + a.m[2][1] * v.z,
                a.m[0][2] * v.x + a.m[1][2] * v.y + a.m[2][2] * v.z);
}
inline mat3 operator*(const mat3& a, float s) {
    mat3 r;
    for (int c = 0; c < 3; c++) for (int k = 0; k < 3; k++) r.m[c][k] = a.m[c][k] * s;
    return r;
}
inline mat3 operator+(const mat3& a, const mat3& b) {
    mat3 r;
    for (int c = 0; c < 3; c++) for (int k = 0; k < 3; k++) r.m[c][k] = a.m[c][k] + b.m[c][k];
    return r;
}
// glm::determinant(mat3)
inline float determinant(const mat3& a) {
    const float(*m)[3] = a.m;
    return +m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2])
           - m[1][0] * (m[0][1] * m[2][2] - m[2][1] * m[0][2])
           + m[2][0] * (m[0][1] * m[1][2] - m[1][1] * m[0][2]);
}
// glm::inverse(mat3) (cofactors * 1/det)
inline mat3 inverse(const mat3& a) {
    const float(*m)[3] = a.m;
    float ood = 1.0f / (+m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2])
                        - m[1][0] * (m[0][1] * m[2][2] - m[2][1] * m[0][2])
                        + m[2][0] * (m[0][1] * m[1][2] - m[1][1] * m[0][2]));
    mat3 r;
    r.m[0][0] = +(m[1][1] * m[2][2] - m[2][1] * m[1][2]) * ood;
    r.m[1][0] = -(m[1][0] * m[2][2] - m[2][0] * m[1][2]) * ood;
    r.m[2][0] = +(m[1][0] * m[2][1] - m[2][0] * m[1][1]) * ood;
    r.m[0][1] = -(m[0][1] * m[2][2] - m[2][1] * m[0][2]) * ood;
    r.m[1][1] = +(m[0][0] * m[2][2] - m[2][0] * m[0][2]) * ood;
    r.m[2][1] = -(m[0][0] * m[2][1] - m[2][0] * m[0][1]) * ood;
    r.m[0][2] = +(m[0][1] * m[1][2] - m[1][1] * m[0][2]) * ood;
    r.m[1][2] = -(m[0][0] * m[1][2] - m[1][0] * m[0][2]) * ood;
    r.m[2][2] = +(m[0][0] * m[1][1] - m[1][0] * m[0][1]) * ood;
    return r;
}

} // namespace orc
