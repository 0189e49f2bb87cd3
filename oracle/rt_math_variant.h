/*
 * TEST INFRASTRUCTURE, oracle only -- never included by the product.
 *
 * A second reading of the GLSL built-ins the render path uses.  The numerics contract (include/rt_math.h) pins one
 * admissible implementation of log / cos / pow / normalize and forbids FMA contraction; a GLSL driver is free to choose
 * others (shaders/raytrace.glsl:202-239 leaves their precision implementation-defined).  This header, force-included
 * ahead of rt_oracle.cpp by `make variant`, swaps in the host C library's logf / cosf / powf, a normalize that divides by
 * the length instead of multiplying by its reciprocal, and is compiled with -ffp-contract=fast (-mfma where the CPU has
 * it): a plausible "other conforming implementation".  tests/test_cpu_oracle.py renders with both and reports how far
 * apart they are -- a bound on what the unpinned part of the contract can move, not a parity claim.
 */
#ifndef RT_MATH_VARIANT_H
#define RT_MATH_VARIANT_H

#include <math.h>

#include "rt_math.h"

static inline rt_vec3 rt_normalize_by_division(rt_vec3 v) {
    float len = sqrtf(rt_dot(v, v));
    return rt_v3(v.x / len, v.y / len, v.z / len);
}
static inline float rt_pow_libm(float x, float y) { return x == 0.0f ? 0.0f : powf(x, y); }

#define rt_log(x) logf(x)
#define rt_cos(x) cosf(x)
#define rt_exp(x) expf(x)
#define rt_pow(x, y) rt_pow_libm(x, y)
#define rt_normalize(v) rt_normalize_by_division(v)

#endif /* RT_MATH_VARIANT_H */
