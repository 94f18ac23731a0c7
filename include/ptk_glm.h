// Minimal stand-in for the handful of glm types the reference's PathTracer API mentions
// (glm::vec2/vec3/vec4/ivec2/mat4, reference PathTracing/src/pathtracer.h:100-130).  When the real glm
// is on the include path (as it is inside the reference application) that is used instead, so
// include/pathtracer.h is source-compatible with the reference's call sites either way.
#pragma once

#if defined(__has_include)
#if __has_include(<glm/glm.hpp>) && !defined(PTK_FORCE_MINI_GLM)
#include <glm/glm.hpp>
#define PTK_HAVE_REAL_GLM 1
#endif
#endif

#ifndef PTK_HAVE_REAL_GLM
namespace glm {

struct vec2 {
    float x, y;
    vec2() : x(0), y(0) {}
    explicit vec2(float s) : x(s), y(s) {}
    vec2(float a, float b) : x(a), y(b) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};

struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}                      // glm 0.9.3.1 zero-initialises (core/type_vec3.inl:67-71)
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};

struct vec4 {
    float x, y, z, w;
    vec4() : x(0), y(0), z(0), w(0) {}
    explicit vec4(float s) : x(s), y(s), z(s), w(s) {}
    vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    vec4(const vec3& v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};

struct ivec2 {
    int x, y;
    ivec2() : x(0), y(0) {}
    ivec2(int a, int b) : x(a), y(b) {}
};

// column-major, m[c][r] like glm
struct mat4 {
    vec4 value[4];
    mat4() : mat4(1.0f) {}
    explicit mat4(float d)
    {
        value[0] = vec4(d, 0, 0, 0); value[1] = vec4(0, d, 0, 0);
        value[2] = vec4(0, 0, d, 0); value[3] = vec4(0, 0, 0, d);
    }
    vec4& operator[](int c) { return value[c]; }
    const vec4& operator[](int c) const { return value[c]; }
};

}  // namespace glm
#endif
