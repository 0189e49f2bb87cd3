/*
 * rt_math.h -- the numerics contract of the render path.
 *
 * The reference's render path is GLSL (shaders/raytrace.glsl), whose built-ins
 * (normalize, log, cos, pow, min, max, ...) have implementation-defined
 * precision and whose expressions a driver may contract into FMAs.  To make
 * "the same pixel draws the same path" a testable, bit-exact statement, every
 * floating-point operation of the path is pinned here, once, as inline
 * functions that compile to the same IEEE-754 binary32 operations with g++
 * (host) and hipcc (gfx950 device):
 *
 *   - only + - * / sqrt, comparisons and integer bit manipulation are used;
 *     / and sqrt are correctly rounded on both sides (x86 SSE; hipcc's default
 *     -fhip-fp32-correctly-rounded-divide-sqrt), denormals are kept;
 *   - every translation unit that includes this header MUST be compiled with
 *     -ffp-contract=off (no FMA contraction) and without fast-math flags;
 *   - operation order follows the GLSL specification's component formulae
 *     (cross, dot, mat3*vec3, mix, smoothstep) evaluated left to right;
 *   - min/max follow fminf/fmaxf: a NaN operand loses.  GLSL leaves that case
 *     undefined (it arises for 0*inf in the slab test, raytrace.glsl:115-120);
 *   - log / cos / exp are fixed polynomial kernels (error <= ~1 ulp for log,
 *     ~1-2 ulp for cos/exp on the ranges the path uses).
 *
 * The header is used by the HIP kernels and by the CPU oracle alike; the
 * tests pin it against libm in double precision.
 */
#ifndef RT_MATH_H
#define RT_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline __attribute__((always_inline))
#endif

/* constants of shaders/raytrace.glsl:7,69,71 */
#define RT_EPS 1e-4f
#define RT_PI 3.1415926f
#define RT_INF (__builtin_inff())

struct rt_vec3 {
    float x, y, z;
};

RT_HD uint32_t rt_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
RT_HD float rt_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

RT_HD rt_vec3 rt_v3(float x, float y, float z) {
    rt_vec3 r;
    r.x = x;
    r.y = y;
    r.z = z;
    return r;
}
RT_HD rt_vec3 rt_add(rt_vec3 a, rt_vec3 b) { return rt_v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD rt_vec3 rt_sub(rt_vec3 a, rt_vec3 b) { return rt_v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD rt_vec3 rt_mul(rt_vec3 a, rt_vec3 b) { return rt_v3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD rt_vec3 rt_scale(rt_vec3 a, float s) { return rt_v3(a.x * s, a.y * s, a.z * s); }
RT_HD rt_vec3 rt_neg(rt_vec3 a) { return rt_v3(-a.x, -a.y, -a.z); }

/* dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z */
RT_HD float rt_dot(rt_vec3 a, rt_vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

/* cross(x,y) = (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y) */
RT_HD rt_vec3 rt_cross(rt_vec3 a, rt_vec3 b) {
    return rt_v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}

RT_HD float rt_sqrt(float x) { return __builtin_sqrtf(x); }
RT_HD float rt_abs(float x) { return __builtin_fabsf(x); }

/* fminf / fmaxf semantics (a NaN operand loses).  The sign of a zero result is
 * not defined by this contract; no consumer on the path observes it. */
RT_HD float rt_min(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fminf(a, b);
#else
    return (a < b || b != b) ? a : b;
#endif
}
RT_HD float rt_max(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmaxf(a, b);
#else
    return (a > b || b != b) ? a : b;
#endif
}

/* normalize(v) = v * inversesqrt(dot(v,v)),  inversesqrt(x) = 1/sqrt(x) */
RT_HD rt_vec3 rt_normalize(rt_vec3 v) {
    float inv = 1.0f / rt_sqrt(rt_dot(v, v));
    return rt_scale(v, inv);
}

/* mat3 * vec3, column-major m[col*3+row] (GLSL / GLM memory order) */
RT_HD rt_vec3 rt_mat3_mul(const float* m, rt_vec3 v) {
    return rt_v3((m[0] * v.x + m[3] * v.y) + m[6] * v.z, (m[1] * v.x + m[4] * v.y) + m[7] * v.z,
                 (m[2] * v.x + m[5] * v.y) + m[8] * v.z);
}

/* natural logarithm.  x = 2^k * m, m in (sqrt(1/2), sqrt(2)], f = m - 1,
 * s = f/(2+f), log(m) = f - f*f/2 + s*(f*f/2 + R(s*s)) with the classic
 * four-term even polynomial R.  log(0) = -inf, log(x<0) = NaN. */
RT_HD float rt_log(float x) {
    uint32_t ix = rt_f2u(x);
    int k = 0;
    if ((ix & 0x7fffffffu) == 0u) return -RT_INF;  /* +-0 */
    if (ix >= 0x80000000u) return rt_u2f(0x7fc00000u); /* negative or -NaN */
    if (ix >= 0x7f800000u) return x;               /* +inf, NaN */
    if (ix < 0x00800000u) {                        /* subnormal: scale by 2^25 (exact) */
        x = x * 33554432.0f;
        ix = rt_f2u(x);
        k = -25;
    }
    k += (int)(ix >> 23) - 127;
    ix = (ix & 0x007fffffu) | 0x3f800000u;         /* m in [1,2) */
    if (ix > 0x3fb504f3u) {                        /* m > sqrt(2) -> m/2 */
        ix -= 0x00800000u;
        k += 1;
    }
    float m = rt_u2f(ix);
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float w = z * z;
    float t1 = w * (0.40000972152f + w * 0.24279078841f);
    float t2 = z * (0.66666662693f + w * 0.28498786688f);
    float R = t2 + t1;
    float hfsq = (0.5f * f) * f;
    float dk = (float)k;
    return dk * 6.9313812256e-01f - ((hfsq - (s * (hfsq + R) + dk * 9.0580006145e-06f)) - f);
}

/* cosine.  |x| is reduced by three-part Cody-Waite subtraction of k*(pi/2),
 * then a degree-7/8 minimax kernel on [-pi/4, pi/4].  Intended for the path's
 * range [0, 2*pi]; stays accurate to a few ulp for |x| < ~1e4. */
RT_HD float rt_cos(float x) {
    float ax = rt_abs(x);
    if (!(ax < 3.0e9f)) {                          /* inf/NaN/huge */
        if (ax != ax || ax == RT_INF) return rt_u2f(0x7fc00000u);
        return 1.0f;                               /* out of supported range */
    }
    uint32_t k = (uint32_t)(ax * 0.63661977237f + 0.5f);
    float fk = (float)k;
    float y = ((ax - fk * 1.5703125f) - fk * 4.837512969970703125e-4f) - fk * 7.549789948768648e-8f;
    float z = y * y;
    float c = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z;
    c = (1.0f - 0.5f * z) + c;
    float sn = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * y + y;
    uint32_t q = k & 3u;
    float r = (q & 1u) ? sn : c;
    return (q == 1u || q == 2u) ? -r : r;
}

/* exponential: k = round(x/ln2), r = x - k*ln2 (two-part), degree-5 kernel,
 * scale by 2^k through the exponent field.  Saturates to 0 / +inf. */
RT_HD float rt_exp(float x) {
    if (x != x) return x;
    if (x > 88.72f) return RT_INF;
    if (x < -87.3f) return 0.0f;                   /* results below FLT_MIN are flushed */
    float t = x * 1.44269504089f;
    int k = (int)(t + (t < 0.0f ? -0.5f : 0.5f));
    float fk = (float)k;
    float r = (x - fk * 0.693359375f) - fk * -2.12194440e-4f;
    float z = r * r;
    float p = ((((1.9875691500e-4f * r + 1.3981999507e-3f) * r + 8.3334519073e-3f) * r + 4.1665795894e-2f) * r +
               1.6666665459e-1f) * r + 5.0000001201e-1f;
    p = (p * z + r) + 1.0f;
    return p * rt_u2f((uint32_t)(k + 127) << 23);
}

/* pow(x,y) = exp(y*log(x)) as GLSL defines it (x >= 0); pow(0, y>0) = 0 */
RT_HD float rt_pow(float x, float y) {
    if (x == 0.0f) return 0.0f;
    return rt_exp(y * rt_log(x));
}

RT_HD float rt_clamp01(float x) { return rt_min(rt_max(x, 0.0f), 1.0f); }

/* smoothstep(e0,e1,x): t = clamp((x-e0)/(e1-e0),0,1); t*t*(3-2t) */
RT_HD float rt_smoothstep(float e0, float e1, float x) {
    float t = rt_clamp01((x - e0) / (e1 - e0));
    return (t * t) * (3.0f - 2.0f * t);
}

/* mix(a,b,t) = a*(1-t) + b*t */
RT_HD float rt_mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }

/* ---- the reference's hash RNG (raytrace.glsl:213-219), exact ---- */
RT_HD uint32_t rt_rand_raw(uint32_t* seed) {
    uint32_t s = *seed * 747796405u + 2891336453u;
    *seed = s;
    uint32_t r = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (r >> 22u) ^ r;
}
/* float(result) / 4294967295.0 : the divisor literal is a float, i.e. 2^32 */
RT_HD float rt_rand(uint32_t* seed) { return (float)rt_rand_raw(seed) * 2.3283064365386962890625e-10f; }

#endif /* RT_MATH_H */
