// Host-side vector/matrix helpers.  Semantics (operation order, constants) follow the
// reference so that prepared scene quantities are bit-identical to Tungsten's:
//   PI = 3.1415926536f (math/Angle.hpp:8), Vec::normalized = v*(1/length) (math/Vec.hpp:168-175),
//   Mat4f products (math/Mat4f.hpp:296-330), rotYXZ (math/Mat4f.cpp:118-130).
#ifndef TGAMD_MATH_HPP_
#define TGAMD_MATH_HPP_

#include <algorithm>
#include <cmath>
#include <cstdint>

namespace tungsten_amd {

static const float PI          = 3.1415926536f;
static const float TWO_PI      = PI*2.0f;
static const float INV_PI      = 1.0f/PI;
static const float INV_TWO_PI  = 0.5f*INV_PI;
static const float INV_FOUR_PI = 0.25f*INV_PI;

struct Vec3f
{
    float v[3];
    Vec3f() : v{0.0f, 0.0f, 0.0f} {}
    explicit Vec3f(float a) : v{a, a, a} {}
    Vec3f(float x, float y, float z) : v{x, y, z} {}
    float x() const { return v[0]; } float y() const { return v[1]; } float z() const { return v[2]; }
    float &operator[](int i) { return v[i]; }
    float operator[](int i) const { return v[i]; }
    Vec3f operator+(const Vec3f &o) const { return Vec3f(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
    Vec3f operator-(const Vec3f &o) const { return Vec3f(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
    Vec3f operator*(const Vec3f &o) const { return Vec3f(v[0]*o.v[0], v[1]*o.v[1], v[2]*o.v[2]); }
    Vec3f operator/(const Vec3f &o) const { return Vec3f(v[0]/o.v[0], v[1]/o.v[1], v[2]/o.v[2]); }
    Vec3f operator*(float s) const { return Vec3f(v[0]*s, v[1]*s, v[2]*s); }
    Vec3f operator/(float s) const { return Vec3f(v[0]/s, v[1]/s, v[2]/s); }
    Vec3f operator-() const { return Vec3f(-v[0], -v[1], -v[2]); }
    Vec3f &operator+=(const Vec3f &o) { for (int i = 0; i < 3; ++i) v[i] += o.v[i]; return *this; }
    Vec3f &operator-=(const Vec3f &o) { for (int i = 0; i < 3; ++i) v[i] -= o.v[i]; return *this; }
    Vec3f &operator*=(float s) { for (int i = 0; i < 3; ++i) v[i] *= s; return *this; }
    Vec3f &operator/=(float s) { for (int i = 0; i < 3; ++i) v[i] /= s; return *this; }
    float dot(const Vec3f &o) const { return v[0]*o.v[0] + v[1]*o.v[1] + v[2]*o.v[2]; }
    Vec3f cross(const Vec3f &o) const
    {
        return Vec3f(v[1]*o.v[2] - v[2]*o.v[1], v[2]*o.v[0] - v[0]*o.v[2], v[0]*o.v[1] - v[1]*o.v[0]);
    }
    float lengthSq() const { return v[0]*v[0] + v[1]*v[1] + v[2]*v[2]; }
    float length() const { return std::sqrt(lengthSq()); }
    Vec3f normalized() const { float inv = 1.0f/length(); return Vec3f(v[0]*inv, v[1]*inv, v[2]*inv); }
    void normalize() { float inv = 1.0f/length(); for (int i = 0; i < 3; ++i) v[i] *= inv; }
    float max() const { return std::max(v[0], std::max(v[1], v[2])); }
    float avg() const { return (v[0] + v[1] + v[2])*(1.0f/3.0f); }
};
inline Vec3f operator*(float s, const Vec3f &a) { return Vec3f(s*a.v[0], s*a.v[1], s*a.v[2]); }
inline Vec3f vmin(const Vec3f &a, const Vec3f &b) { return Vec3f(std::min(a[0], b[0]), std::min(a[1], b[1]), std::min(a[2], b[2])); }
inline Vec3f vmax(const Vec3f &a, const Vec3f &b) { return Vec3f(std::max(a[0], b[0]), std::max(a[1], b[1]), std::max(a[2], b[2])); }

struct Box3f
{
    Vec3f lo, hi;
    Box3f() : lo(1e30f), hi(-1e30f) {}
    void grow(const Vec3f &p) { lo = vmin(lo, p); hi = vmax(hi, p); }
    void grow(const Box3f &b) { lo = vmin(lo, b.lo); hi = vmax(hi, b.hi); }
    bool empty() const { return lo[0] > hi[0]; }
    float area() const { Vec3f d = hi - lo; return 2.0f*(d[0]*d[1] + d[1]*d[2] + d[2]*d[0]); }
};

struct Mat4f
{
    float a[16];
    Mat4f() { for (int i = 0; i < 16; ++i) a[i] = (i % 5 == 0) ? 1.0f : 0.0f; }
    Mat4f(const Vec3f &right, const Vec3f &up, const Vec3f &fwd) : Mat4f()
    {
        a[0] = right[0]; a[1] = up[0]; a[2]  = fwd[0];
        a[4] = right[1]; a[5] = up[1]; a[6]  = fwd[1];
        a[8] = right[2]; a[9] = up[2]; a[10] = fwd[2];
    }
    float operator[](int i) const { return a[i]; }
    float &operator[](int i) { return a[i]; }
    // math/Mat4f.hpp:75-107: cofactor expansion, then one multiplication by 1/det
    Mat4f invert() const
    {
        Mat4f inv;
        inv[ 0] =  a[5]*a[10]*a[15] - a[5]*a[11]*a[14] - a[9]*a[6]*a[15] + a[9]*a[7]*a[14] + a[13]*a[6]*a[11] - a[13]*a[7]*a[10];
        inv[ 1] = -a[1]*a[10]*a[15] + a[1]*a[11]*a[14] + a[9]*a[2]*a[15] - a[9]*a[3]*a[14] - a[13]*a[2]*a[11] + a[13]*a[3]*a[10];
        inv[ 2] =  a[1]*a[ 6]*a[15] - a[1]*a[ 7]*a[14] - a[5]*a[2]*a[15] + a[5]*a[3]*a[14] + a[13]*a[2]*a[ 7] - a[13]*a[3]*a[ 6];
        inv[ 3] = -a[1]*a[ 6]*a[11] + a[1]*a[ 7]*a[10] + a[5]*a[2]*a[11] - a[5]*a[3]*a[10] - a[ 9]*a[2]*a[ 7] + a[ 9]*a[3]*a[ 6];
        inv[ 4] = -a[4]*a[10]*a[15] + a[4]*a[11]*a[14] + a[8]*a[6]*a[15] - a[8]*a[7]*a[14] - a[12]*a[6]*a[11] + a[12]*a[7]*a[10];
        inv[ 5] =  a[0]*a[10]*a[15] - a[0]*a[11]*a[14] - a[8]*a[2]*a[15] + a[8]*a[3]*a[14] + a[12]*a[2]*a[11] - a[12]*a[3]*a[10];
        inv[ 6] = -a[0]*a[ 6]*a[15] + a[0]*a[ 7]*a[14] + a[4]*a[2]*a[15] - a[4]*a[3]*a[14] - a[12]*a[2]*a[ 7] + a[12]*a[3]*a[ 6];
        inv[ 8] =  a[4]*a[ 9]*a[15] - a[4]*a[11]*a[13] - a[8]*a[5]*a[15] + a[8]*a[7]*a[13] + a[12]*a[5]*a[11] - a[12]*a[7]*a[ 9];
        inv[ 7] =  a[0]*a[ 6]*a[11] - a[0]*a[ 7]*a[10] - a[4]*a[2]*a[11] + a[4]*a[3]*a[10] + a[ 8]*a[2]*a[ 7] - a[ 8]*a[3]*a[ 6];
        inv[ 9] = -a[0]*a[ 9]*a[15] + a[0]*a[11]*a[13] + a[8]*a[1]*a[15] - a[8]*a[3]*a[13] - a[12]*a[1]*a[11] + a[12]*a[3]*a[ 9];
        inv[10] =  a[0]*a[ 5]*a[15] - a[0]*a[ 7]*a[13] - a[4]*a[1]*a[15] + a[4]*a[3]*a[13] + a[12]*a[1]*a[ 7] - a[12]*a[3]*a[ 5];
        inv[11] = -a[0]*a[ 5]*a[11] + a[0]*a[ 7]*a[ 9] + a[4]*a[1]*a[11] - a[4]*a[3]*a[ 9] - a[ 8]*a[1]*a[ 7] + a[ 8]*a[3]*a[ 5];
        inv[12] = -a[4]*a[ 9]*a[14] + a[4]*a[10]*a[13] + a[8]*a[5]*a[14] - a[8]*a[6]*a[13] - a[12]*a[5]*a[10] + a[12]*a[6]*a[ 9];
        inv[13] =  a[0]*a[ 9]*a[14] - a[0]*a[10]*a[13] - a[8]*a[1]*a[14] + a[8]*a[2]*a[13] + a[12]*a[1]*a[10] - a[12]*a[2]*a[ 9];
        inv[14] = -a[0]*a[ 5]*a[14] + a[0]*a[ 6]*a[13] + a[4]*a[1]*a[14] - a[4]*a[2]*a[13] - a[12]*a[1]*a[ 6] + a[12]*a[2]*a[ 5];
        inv[15] =  a[0]*a[ 5]*a[10] - a[0]*a[ 6]*a[ 9] - a[4]*a[1]*a[10] + a[4]*a[2]*a[ 9] + a[ 8]*a[1]*a[ 6] - a[ 8]*a[2]*a[ 5];
        float det = a[0]*inv[0] + a[1]*inv[4] + a[2]*inv[8] + a[3]*inv[12];
        if (det == 0.0f)
            return Mat4f();
        float invDet = 1.0f/det;
        for (int i = 0; i < 16; ++i) inv[i] = inv[i]*invDet;
        return inv;
    }
    Vec3f right() const { return Vec3f(a[0], a[4], a[8]); }
    Vec3f up()    const { return Vec3f(a[1], a[5], a[9]); }
    Vec3f fwd()   const { return Vec3f(a[2], a[6], a[10]); }
    void setRight(const Vec3f &x) { a[0] = x[0]; a[4] = x[1]; a[8] = x[2]; }
    Vec3f translation() const { return Vec3f(a[3], a[7], a[11]); }
    Vec3f transformVector(const Vec3f &b) const
    {
        return Vec3f(
            a[0]*b[0] + a[1]*b[1] + a[2]*b[2],
            a[4]*b[0] + a[5]*b[1] + a[6]*b[2],
            a[8]*b[0] + a[9]*b[1] + a[10]*b[2]);
    }
    Vec3f operator*(const Vec3f &b) const
    {
        return Vec3f(
            a[0]*b[0] + a[1]*b[1] + a[2]*b[2]  + a[3],
            a[4]*b[0] + a[5]*b[1] + a[6]*b[2]  + a[7],
            a[8]*b[0] + a[9]*b[1] + a[10]*b[2] + a[11]);
    }
    Mat4f operator*(const Mat4f &b) const
    {
        Mat4f r;
        for (int i = 0; i < 4; i++)
            for (int t = 0; t < 4; t++)
                r.a[i*4 + t] = a[i*4 + 0]*b.a[0*4 + t] + a[i*4 + 1]*b.a[1*4 + t] + a[i*4 + 2]*b.a[2*4 + t] + a[i*4 + 3]*b.a[3*4 + t];
        return r;
    }
    Mat4f transpose() const
    {
        Mat4f r;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.a[i*4 + j] = a[j*4 + i];
        return r;
    }
    static Mat4f scale(const Vec3f &s) { Mat4f r; r.a[0] = s[0]; r.a[5] = s[1]; r.a[10] = s[2]; return r; }
    // math/Mat4f.cpp:40-47, :54-57, :10-13
    Mat4f extractRotation() const { return Mat4f(right().normalized(), up().normalized(), fwd().normalized()); }
    Vec3f extractScaleVec() const { return Vec3f(right().length(), up().length(), fwd().length()); }
    Mat4f toNormalMatrix() const
    {
        return scale(Vec3f(1.0f)/Vec3f(right().lengthSq(), up().lengthSq(), fwd().lengthSq()))*(*this);
    }
    // math/Mat4f.cpp:118-130
    static Mat4f rotYXZ(const Vec3f &rot)
    {
        Vec3f r = rot*PI/180.0f;
        float c[] = {std::cos(r[0]), std::cos(r[1]), std::cos(r[2])};
        float s[] = {std::sin(r[0]), std::sin(r[1]), std::sin(r[2])};
        Mat4f m;
        m.a[0] = c[1]*c[2] - s[1]*s[0]*s[2]; m.a[1] = -c[1]*s[2] - s[1]*s[0]*c[2]; m.a[2]  = -s[1]*c[0];
        m.a[4] = c[0]*s[2];                  m.a[5] = c[0]*c[2];                   m.a[6]  = -s[0];
        m.a[8] = s[1]*c[2] + c[1]*s[0]*s[2]; m.a[9] = -s[1]*s[2] + c[1]*s[0]*c[2]; m.a[10] = c[1]*c[0];
        return m;
    }
};

// math/Quaternion.hpp:13-170 (w, x, y, z), the operations Instance uses
struct QuaternionF
{
    float v[4];
    QuaternionF() { v[0] = 1.0f; v[1] = v[2] = v[3] = 0.0f; }
    QuaternionF(float w, float x, float y, float z) { v[0] = w; v[1] = x; v[2] = y; v[3] = z; }
    QuaternionF(float theta, const Vec3f &u)                       // :32-40
    {
        float cosTheta = std::cos(theta/2.0f), sinTheta = std::sin(theta/2.0f);
        v[0] = cosTheta; v[1] = u[0]*sinTheta; v[2] = u[1]*sinTheta; v[3] = u[2]*sinTheta;
    }
    float operator[](int i) const { return v[i]; }
    QuaternionF conjugate() const { return QuaternionF(v[0], -v[1], -v[2], -v[3]); }
    QuaternionF operator*(const QuaternionF &o) const              // :68-76
    {
        return QuaternionF(
            v[0]*o[0] - v[1]*o[1] - v[2]*o[2] - v[3]*o[3],
            v[0]*o[1] + v[1]*o[0] + v[2]*o[3] - v[3]*o[2],
            v[0]*o[2] - v[1]*o[3] + v[2]*o[0] + v[3]*o[1],
            v[0]*o[3] + v[1]*o[2] - v[2]*o[1] + v[3]*o[0]);
    }
    Vec3f operator*(const Vec3f &o) const                          // :78-88
    {
        float tx = 2.0f*(v[2]*o[2] - v[3]*o[1]);
        float ty = 2.0f*(v[3]*o[0] - v[1]*o[2]);
        float tz = 2.0f*(v[1]*o[1] - v[2]*o[0]);
        return Vec3f(
            o[0] + v[0]*tx + v[2]*tz - v[3]*ty,
            o[1] + v[0]*ty + v[3]*tx - v[1]*tz,
            o[2] + v[0]*tz + v[1]*ty - v[2]*tx);
    }
    static QuaternionF fromMatrix(const Mat4f &m)                  // :112-150; a(i, j) = row i, column j
    {
        auto a = [&m](int i, int j) { return m[i*4 + j]; };
        float trace = a(0, 0) + a(1, 1) + a(2, 2);
        if (trace > 0.0f) {
            float s = 0.5f/std::sqrt(trace + 1.0f);
            return QuaternionF(0.25f/s, (a(2, 1) - a(1, 2))*s, (a(0, 2) - a(2, 0))*s, (a(1, 0) - a(0, 1))*s);
        } else if (a(0, 0) > a(1, 1) && a(0, 0) > a(2, 2)) {
            float s = 2.0f*std::sqrt(1.0f + a(0, 0) - a(1, 1) - a(2, 2));
            return QuaternionF((a(2, 1) - a(1, 2))/s, 0.25f*s, (a(0, 1) + a(1, 0))/s, (a(0, 2) + a(2, 0))/s);
        } else if (a(1, 1) > a(2, 2)) {
            float s = 2.0f*std::sqrt(1.0f + a(1, 1) - a(0, 0) - a(2, 2));
            return QuaternionF((a(0, 2) - a(2, 0))/s, (a(0, 1) + a(1, 0))/s, 0.25f*s, (a(1, 2) + a(2, 1))/s);
        } else {
            float s = 2.0f*std::sqrt(1.0f + a(2, 2) - a(0, 0) - a(1, 1));
            return QuaternionF((a(1, 0) - a(0, 1))/s, (a(0, 2) + a(2, 0))/s, (a(1, 2) + a(2, 1))/s, 0.25f*s);
        }
    }
};

// math/MathUtil.hpp:120-128
static inline uint32_t hash32(uint32_t x)
{
    x = ~x + (x << 15);
    x = x ^ (x >> 12);
    x = x + (x << 2);
    x = x ^ (x >> 4);
    x = x * 2057;
    x = x ^ (x >> 16);
    return x;
}

} // namespace tungsten_amd

#endif
