// sfl/vector.h -- element types of the fluid fields, source compatible with the reference
// sketch's vector.h (ESP32-fluid-simulation/vector.h:4-61 Vector2, :63-126 Vector3):
// same names, same public members x / y (/ z), same constructors, same operator set and the
// same promotion rule (multiplying or dividing by a float always yields Vector<float>,
// vector.h:51-56,116-121).  Layout is the plain aggregate {x, y} / {x, y, z}: Vector2<float> is
// 8 bytes, Vector3<UQ32> 12 bytes -- exactly what the device kernels read (include/sfl.h).
//
// Written from scratch for this library; everything is usable from host and HIP device code.
#ifndef SFL_VECTOR_H
#define SFL_VECTOR_H

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SFL_XPU __host__ __device__
#else
#define SFL_XPU
#endif

template <typename T>
struct Vector2 {
    T x, y;

    SFL_XPU Vector2() {}
    template <typename U>
    SFL_XPU Vector2(U x_, U y_) : x(x_), y(y_) {}
    template <typename U>
    SFL_XPU Vector2(Vector2<U> other) : x(other.x), y(other.y) {}

    SFL_XPU Vector2 &operator=(const Vector2 &o) { x = o.x; y = o.y; return *this; }
    SFL_XPU Vector2 &operator+=(const Vector2 &o) { x += o.x; y += o.y; return *this; }
    SFL_XPU Vector2 &operator-=(const Vector2 &o) { x -= o.x; y -= o.y; return *this; }
    SFL_XPU Vector2 &operator*=(const float &s) { x *= s; y *= s; return *this; }
    SFL_XPU Vector2 &operator/=(const float &s) { x /= s; y /= s; return *this; }

    // members (not templates) so that the right-hand side converts implicitly, as in the reference
    SFL_XPU Vector2 operator-() const { return Vector2(-x, -y); }
    SFL_XPU Vector2 operator+(const Vector2 &o) const { return Vector2(x + o.x, y + o.y); }
    SFL_XPU Vector2 operator-(const Vector2 &o) const { return Vector2(x - o.x, y - o.y); }
};

template <typename T>
struct Vector3 {
    T x, y, z;

    SFL_XPU Vector3() {}
    template <typename U>
    SFL_XPU Vector3(U x_, U y_, U z_) : x(x_), y(y_), z(z_) {}
    template <typename U>
    SFL_XPU Vector3(Vector3<U> other) : x(other.x), y(other.y), z(other.z) {}

    SFL_XPU Vector3 &operator=(const Vector3 &o) { x = o.x; y = o.y; z = o.z; return *this; }
    SFL_XPU Vector3 &operator+=(const Vector3 &o) { x += o.x; y += o.y; z += o.z; return *this; }
    SFL_XPU Vector3 &operator-=(const Vector3 &o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
    SFL_XPU Vector3 &operator*=(const float &s) { x *= s; y *= s; z *= s; return *this; }
    SFL_XPU Vector3 &operator/=(const float &s) { x /= s; y /= s; z /= s; return *this; }

    SFL_XPU Vector3 operator-() const { return Vector3(-x, -y, -z); }
    SFL_XPU Vector3 operator+(const Vector3 &o) const { return Vector3(x + o.x, y + o.y, z + o.z); }
    SFL_XPU Vector3 operator-(const Vector3 &o) const { return Vector3(x - o.x, y - o.y, z - o.z); }
};

// ---- scaling promotes to float components (each product / quotient rounded on its own) ----
template <typename T> SFL_XPU inline Vector2<float> operator*(const Vector2<T> &a, const float &s) { return Vector2<float>(a.x * s, a.y * s); }
template <typename T> SFL_XPU inline Vector2<float> operator/(const Vector2<T> &a, const float &s) { return Vector2<float>(a.x / s, a.y / s); }
template <typename T> SFL_XPU inline Vector2<float> operator*(const float &s, const Vector2<T> &a) { return a * s; }
template <typename T> SFL_XPU inline Vector3<float> operator*(const Vector3<T> &a, const float &s) { return Vector3<float>(a.x * s, a.y * s, a.z * s); }
template <typename T> SFL_XPU inline Vector3<float> operator/(const Vector3<T> &a, const float &s) { return Vector3<float>(a.x / s, a.y / s, a.z / s); }
template <typename T> SFL_XPU inline Vector3<float> operator*(const float &s, const Vector3<T> &a) { return a * s; }

#endif  // SFL_VECTOR_H
