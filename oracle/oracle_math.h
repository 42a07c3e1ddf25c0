/*
 * oracle_math.h — the ORACLE's own statement of the fp32 geometry of the map update (test infrastructure, like everything under
 * oracle/; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it).
 *
 * Until round 5 the oracle included the product's include/gie_math.h, so a wrong sign or rounding there was common to both sides
 * of every parity test.  This header shares no line with it: written from the reference's expressions
 * (cited per function), in a different spelling wherever IEEE-754 allows one that rounds identically.  What must be identical is
 * the VALUE after every fp32 rounding — voxelisation floors it — and tests/test_independent_checks.py holds the two headers
 * against each other bit for bit on random inputs, and each against float64.  Compile with -ffp-contract=off.
 */
#ifndef ORACLE_MATH_H
#define ORACLE_MATH_H

#include <math.h>

typedef struct om_pose {
    float r[3][3];      /* rotation, row-major */
    float t[3];         /* translation */
} om_pose;

/* SE3 from a unit quaternion and a translation (/root/reference/include/cuda_toolkit/se3.cuh:47-77): the textbook matrix with the
 * quaternion's components doubled FIRST (2x, 2y, 2z), then multiplied — fl((2a) b) = 2 fl(a b), the same value as the sum of two
 * equal products */
static inline om_pose om_from_quat(const float q[4], const float t[3])
{
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float x2 = x + x, y2 = y + y, z2 = z + z;
    const float xx = x2 * x, yy = y2 * y, zz = z2 * z;
    const float xy = x2 * y, xz = x2 * z, yz = y2 * z;
    const float wx = x2 * w, wy = y2 * w, wz = z2 * w;
    om_pose p;
    p.r[0][0] = 1.0f - (yy + zz); p.r[0][1] = xy - wz;          p.r[0][2] = xz + wy;
    p.r[1][0] = xy + wz;          p.r[1][1] = 1.0f - (xx + zz); p.r[1][2] = yz - wx;
    p.r[2][0] = xz - wy;          p.r[2][1] = yz + wx;          p.r[2][2] = 1.0f - (xx + yy);
    p.t[0] = t[0]; p.t[1] = t[1]; p.t[2] = t[2];
    return p;
}

/* the rigid inverse (se3.cuh:91-108): R^T and -R^T t, column c of R against t, the products taken away one after the other */
static inline om_pose om_inverse(const om_pose a)
{
    om_pose b;
    for (int c = 0; c < 3; c++) {
        for (int k = 0; k < 3; k++) b.r[c][k] = a.r[k][c];
        float acc = -(a.r[0][c] * a.t[0]);
        acc = acc - a.r[1][c] * a.t[1];
        acc = acc - a.r[2][c] * a.t[2];
        b.t[c] = acc;
    }
    return b;
}

/* rotate, then translate (se3.cuh:123-149, 200-204): the three products of a row summed from the left */
static inline void om_transform(const om_pose a, const float p[3], float out[3])
{
    for (int i = 0; i < 3; i++) {
        float s = a.r[i][0] * p[0];
        s = s + a.r[i][1] * p[1];
        s = s + a.r[i][2] * p[2];
        out[i] = s + a.t[i];
    }
}

/* position -> voxel index (/root/reference/include/map_structure/local_batch.h:250-258): floor(p / w + 1/2) */
static inline int om_voxel_of(float p, float w)
{
    const float q = p / w;
    return (int)floorf(q + 0.5f);
}

/* a cloud point that can be voxelised at all: every coordinate finite and within a million metres (NaN fails both comparisons).
 * The reference converts whatever it is given (pntcld_raycast.cu:88-94): undefined for such points; both sides of the parity
 * tests ignore them (include/gie.h, gie_ogm_pointcloud). */
static inline int om_point_usable(const float g[3])
{
    for (int i = 0; i < 3; i++) if (!(g[i] >= -1.0e6f && g[i] <= 1.0e6f)) return 0;
    return 1;
}

/* atan2 as BOTH sides of the parity tests define it (the reference calls CUDA's atan2f under -use_fast_math, which cannot be
 * reproduced off that platform): arctangent of |y| / |x| by the three-interval reduction at tan(pi/8), tan(3 pi/8) and the odd
 * polynomial of degree 9 with the coefficients below, then the quadrant.  The value is pinned against libm to 2 ulp in
 * tests/test_independent_checks.py. */
static inline float om_arctan_first_quadrant(float r)
{
    static const float c9 = 8.05374449538e-2f, c7 = -1.38776856032e-1f, c5 = 1.99777106478e-1f, c3 = -3.33329491539e-1f;
    float base = 0.0f, u = r;
    if (r > 2.414213562373095f) { base = 1.5707963267948966f; u = -(1.0f / r); }
    else if (r > 0.4142135623730950f) { base = 0.7853981633974483f; u = (r - 1.0f) / (r + 1.0f); }
    const float s = u * u;
    float h = c9 * s + c7;
    h = h * s + c5;
    h = h * s + c3;
    const float tail = h * s * u;
    return base + (tail + u);
}
static inline float om_atan2(float y, float x)
{
    const float pi = 3.14159265358979323846f, half_pi = 1.57079632679489661923f;
    if (x == 0.0f) return y > 0.0f ? half_pi : (y < 0.0f ? -half_pi : 0.0f);
    if (y == 0.0f) return x > 0.0f ? 0.0f : pi;
    const float a = om_arctan_first_quadrant(fabsf(y) / fabsf(x));
    if (x > 0.0f) return y > 0.0f ? a : -a;
    return y > 0.0f ? pi - a : a - pi;
}

#endif /* ORACLE_MATH_H */
