// TEST INFRASTRUCTURE — CPU oracle. Not part of the shipped library; only tests,
// the smoke check and the CPU-baseline leg of the benchmark may link this.
//
// Restatement of the nalgebra 0.33.0 arithmetic the reference calls
// (Cargo.lock:2287-2288; nalgebra itself is not vendored under the reference).
// What matters here is the ORDER of f32 operations, because parity is judged in
// bits for intersection results:
//   dot (3)        (a0*b0 + a1*b1) + a2*b2
//   cross          textbook component form
//   norm_squared   dot(v, v)
//   normalize      component-wise DIVIDE by sqrt(norm_squared)
//   M * v          column by column: y = col0*x0; y = col1*x1 + y; y = col2*x2 + y
//   M4 * M4        the same accumulation per output column
//   lerp(a, b, t)  t*b + (1-t)*a
//   center(a, b)   (a + b) * 0.5
//   transform_point on Matrix4: (M3*p + translation) / (row3 . p + m33) when non-zero
//   transform_vector on Matrix4: M3*v (divided by row3 . v when non-zero)
//   Matrix4::new_rotation(axis*angle): Rodrigues form of Rotation3::from_axis_angle
// parity unpinned: the reference ships no vectors for any of these; call sites are
// cited next to each use.
#pragma once
#include <cmath>
#include <cstdint>

namespace oracle {

struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float &at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};

inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 cmul(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float norm_squared(V3 a) { return dot(a, a); }
inline float norm(V3 a) { return std::sqrt(norm_squared(a)); }
inline V3 normalize(V3 a) {
    float n = norm(a);
    return {a.x / n, a.y / n, a.z / n};
}
inline V3 lerp(V3 a, V3 b, float t) {
    float s = 1.0f - t;
    return {t * b.x + s * a.x, t * b.y + s * a.y, t * b.z + s * a.z};
}
inline V3 center(V3 a, V3 b) { return (a + b) * 0.5f; }

// Rust f32::min / f32::max: a NaN operand is ignored.
inline float rmin(float a, float b) { return std::fmin(a, b); }
inline float rmax(float a, float b) { return std::fmax(a, b); }
// Rust f32::clamp: NaN stays NaN.
inline float rclamp(float x, float lo, float hi) {
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}

// 3x3 with columns c0,c1,c2 (Matrix3::from_columns).
struct M3 {
    V3 c0, c1, c2;
};
inline V3 mul(const M3 &m, V3 v) {
    V3 y = m.c0 * v.x;
    y = m.c1 * v.y + y;
    y = m.c2 * v.z + y;
    return y;
}
inline V3 mul_transpose(const M3 &m, V3 v) {
    // transpose() materialises the matrix, then the same column accumulation runs on
    // it; row i of the result is ((c_i.x*v.x) + c_i.y*v.y) + c_i.z*v.z.
    return {dot(m.c0, v), dot(m.c1, v), dot(m.c2, v)};
}

// 4x4, m[row][col].
struct M4 {
    float m[4][4];
};
inline M4 m4_identity() {
    M4 r{};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r.m[i][j] = (i == j) ? 1.0f : 0.0f;
    return r;
}
inline M4 mul(const M4 &a, const M4 &b) {
    M4 r;
    for (int j = 0; j < 4; j++) {
        float y[4];
        for (int i = 0; i < 4; i++) y[i] = a.m[i][0] * b.m[0][j];
        for (int k = 1; k < 4; k++)
            for (int i = 0; i < 4; i++) y[i] = a.m[i][k] * b.m[k][j] + y[i];
        for (int i = 0; i < 4; i++) r.m[i][j] = y[i];
    }
    return r;
}
inline M4 m4_translation(V3 t) {
    M4 r = m4_identity();
    r.m[0][3] = t.x;
    r.m[1][3] = t.y;
    r.m[2][3] = t.z;
    return r;
}
inline M4 m4_scaling(V3 s) {
    M4 r = m4_identity();
    r.m[0][0] = s.x;
    r.m[1][1] = s.y;
    r.m[2][2] = s.z;
    return r;
}
// Matrix4::new_rotation(axisangle) = Rotation3::new(axisangle).to_homogeneous().
inline M4 m4_rotation(V3 axisangle) {
    float angle = norm(axisangle);
    M4 r = m4_identity();
    if (angle != 0.0f) {
        V3 u = axisangle / angle;
        float sqx = u.x * u.x, sqy = u.y * u.y, sqz = u.z * u.z;
        float s = std::sin(angle), c = std::cos(angle);
        float omc = 1.0f - c;
        r.m[0][0] = sqx + (1.0f - sqx) * c;
        r.m[0][1] = u.x * u.y * omc - u.z * s;
        r.m[0][2] = u.x * u.z * omc + u.y * s;
        r.m[1][0] = u.x * u.y * omc + u.z * s;
        r.m[1][1] = sqy + (1.0f - sqy) * c;
        r.m[1][2] = u.y * u.z * omc - u.x * s;
        r.m[2][0] = u.x * u.z * omc - u.y * s;
        r.m[2][1] = u.y * u.z * omc + u.x * s;
        r.m[2][2] = sqz + (1.0f - sqz) * c;
    }
    return r;
}
inline V3 m4_mul3(const M4 &m, V3 v) {
    V3 y = v3(m.m[0][0], m.m[1][0], m.m[2][0]) * v.x;
    y = v3(m.m[0][1], m.m[1][1], m.m[2][1]) * v.y + y;
    y = v3(m.m[0][2], m.m[1][2], m.m[2][2]) * v.z + y;
    return y;
}
inline V3 transform_point(const M4 &m, V3 p) {
    float n = dot(v3(m.m[3][0], m.m[3][1], m.m[3][2]), p) + m.m[3][3];
    V3 r = m4_mul3(m, p) + v3(m.m[0][3], m.m[1][3], m.m[2][3]);
    if (n != 0.0f) return r / n;
    return r;
}
inline V3 transform_vector(const M4 &m, V3 v) {
    float n = dot(v3(m.m[3][0], m.m[3][1], m.m[3][2]), v);
    if (n != 0.0f) return m4_mul3(m, v / n);
    return m4_mul3(m, v);
}

constexpr float PI_F = 3.14159265358979323846f;

}  // namespace oracle
