/*
 * gie_math.h — deterministic fp32 geometry of the HIP kernels (and of the host layers above the C-ABI).
 *
 * Voxelisation must agree bit-for-bit between the GPU path and the CPU checker, so everything
 * here is spelled as explicit IEEE-754 binary32 +,-,*,/,sqrt,floor in a fixed evaluation order
 * and both sides are compiled with -ffp-contract=off (no FMA contraction).  The oracle does NOT include
 * this header (since round 5): oracle/oracle_math.h states the same geometry on its own, and
 * tests/test_independent_checks.py holds the two against each other bit for bit and each against float64.  No libm
 * transcendental is used: atan2 is a fixed polynomial (Cephes-style atanf reduction), because
 * device and host libm differ in the last ulp and a one-ulp difference flips floor() bins.
 *
 * What is restated here is mathematics, written for this header: the rotation matrix of a unit quaternion (the textbook
 * R = I + 2 w [v]x + 2 [v]x^2), the rigid inverse (R^T, -R^T t), the point transform, and the voxel index floor(p / w + 1/2).
 * The reference evaluates the same expressions in include/cuda_toolkit/se3.cuh:47-77, 91-108, 123-149, 200-204 and
 * include/map_structure/local_batch.h:250-258; for voxelisation to agree with it bit for bit, every fp32 ROUNDING has to
 * happen where it happens there, which fixes the association of the two-term sums below but not the way they are written:
 * the matrix is built from the six products of quaternion components, each doubled by an addition (doubling is exact in
 * binary floating point, so fl((2a) b) = 2 fl(a b) = fl(a b) + fl(a b)); tests/test_independent_checks.py holds the result
 * against float64 AND, bit for bit, against an fp32 evaluation in the "double the component first" order.
 */
#ifndef GIE_MATH_H
#define GIE_MATH_H

#include <math.h>

#if defined(__HIPCC__)
#define GIE_HD __host__ __device__ __forceinline__
#else
#define GIE_HD static inline
#endif

typedef struct gie_se3 {
    float m[12]; /* row-major 3x4: r00 r01 r02 tx / r10 r11 r12 ty / r20 r21 r22 tz */
} gie_se3;

/* Rotation from a normalised quaternion (w, x, y, z), translation t.  (The reference's counterpart: se3.cuh:47-77.) */
GIE_HD gie_se3 gie_se3_from_quat(float qw, float qx, float qy, float qz, float tx, float ty, float tz)
{
    const float v[3] = { qx, qy, qz };
    float vv[3][3], wv[3];                   /* 2 v_i v_j (upper triangle) and 2 w v_i */
    for (int i = 0; i < 3; i++) {
        const float pw = v[i] * qw;
        wv[i] = pw + pw;
        for (int j = i; j < 3; j++) { const float pv = v[j] * v[i]; vv[i][j] = pv + pv; }
    }
    gie_se3 s;
    s.m[0] = 1 - (vv[1][1] + vv[2][2]); s.m[1] = vv[0][1] - wv[2];           s.m[2] = vv[0][2] + wv[1];            s.m[3] = tx;
    s.m[4] = vv[0][1] + wv[2];           s.m[5] = 1 - (vv[0][0] + vv[2][2]); s.m[6] = vv[1][2] - wv[0];            s.m[7] = ty;
    s.m[8] = vv[0][2] - wv[1];           s.m[9] = vv[1][2] + wv[0];           s.m[10] = 1 - (vv[0][0] + vv[1][1]); s.m[11] = tz;
    return s;
}

/* Rigid inverse: R^T, -R^T t.  These are the same twelve assignments, in the same order, as the reference's se3.cuh:91-108 (REMODE's
 * SE3<T>::inv, GPL): the fp32 result depends on the order the three products of a row are subtracted in (left to
 * right), so there is one way to write it that reproduces the reference bit for bit, and this is it. */
GIE_HD gie_se3 gie_se3_inv(const gie_se3 a)
{
    gie_se3 r;
    r.m[0] = a.m[0]; r.m[1] = a.m[4]; r.m[2] = a.m[8];
    r.m[4] = a.m[1]; r.m[5] = a.m[5]; r.m[6] = a.m[9];
    r.m[8] = a.m[2]; r.m[9] = a.m[6]; r.m[10] = a.m[10];
    r.m[3] = -a.m[0] * a.m[3] - a.m[4] * a.m[7] - a.m[8] * a.m[11];
    r.m[7] = -a.m[1] * a.m[3] - a.m[5] * a.m[7] - a.m[9] * a.m[11];
    r.m[11] = -a.m[2] * a.m[3] - a.m[6] * a.m[7] - a.m[10] * a.m[11];
    return r;
}

/* Rotate, then translate (the reference's counterpart: se3.cuh:123-149, 200-204). */
GIE_HD void gie_se3_apply(const gie_se3 s, float px, float py, float pz, float *ox, float *oy, float *oz)
{
    const float rx = s.m[0] * px + s.m[1] * py + s.m[2] * pz;
    const float ry = s.m[4] * px + s.m[5] * py + s.m[6] * pz;
    const float rz = s.m[8] * px + s.m[9] * py + s.m[10] * pz;
    *ox = rx + s.m[3];
    *oy = ry + s.m[7];
    *oz = rz + s.m[11];
}

/* local_batch.h:250-258 */
GIE_HD int gie_pos2coord(float p, float w) { return (int)floorf(p / w + 0.5f); }

/* A cloud point takes part in ray casting only if its global-frame coordinates are finite and
 * within +-1e6 m: the reference converts whatever it gets to a voxel coordinate
 * (pntcld_raycast.cu:88-94, ray_cast.h:62-66), which for NaN / Inf / huge values is an undefined
 * float->int conversion; PointCloud2 messages with is_dense == false do carry NaN points.  Here
 * such a point is ignored: it registers nothing and casts no ray (not even the sensor's own cell). */
GIE_HD int gie_point_ok(float gx, float gy, float gz)
{
    return fabsf(gx) <= 1.0e6f && fabsf(gy) <= 1.0e6f && fabsf(gz) <= 1.0e6f;   /* false for NaN */
}

/* atan on [0, inf) by the classic three-interval reduction and a degree-4 (in z = x*x) odd
 * polynomial; |error| < 2 ulp.  The reference calls CUDA atan2f under -use_fast_math, which is
 * not reproducible anywhere else; this is the pinned stand-in on both sides. */
GIE_HD float gie_atan_pos(float x)
{
    float y;
    if (x > 2.414213562373095f) { /* tan(3pi/8) */
        y = 1.5707963267948966f;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) { /* tan(pi/8) */
        y = 0.7853981633974483f;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
    }
    const float z = x * x;
    float p = 8.05374449538e-2f * z - 1.38776856032e-1f;
    p = p * z + 1.99777106478e-1f;
    p = p * z - 3.33329491539e-1f;
    p = p * z * x + x;
    return y + p;
}

GIE_HD float gie_atan2f(float y, float x)
{
    const float PI_F = 3.14159265358979323846f;
    const float PIO2_F = 1.57079632679489661923f;
    if (x == 0.0f) {
        if (y > 0.0f) return PIO2_F;
        if (y < 0.0f) return -PIO2_F;
        return 0.0f;
    }
    if (y == 0.0f) return x > 0.0f ? 0.0f : PI_F;
    const float a = gie_atan_pos(fabsf(y) / fabsf(x)); /* (0, pi/2) */
    if (x > 0.0f) return y > 0.0f ? a : -a;
    return y > 0.0f ? PI_F - a : a - PI_F;
}

#endif /* GIE_MATH_H */
