/* TEST-ONLY: exports include/gie_math.h (the fp32 geometry shared by the HIP kernels and the oracle) so that
 * tests/test_independent_checks.py can pin it against double precision and hand-derived vectors.
 *   gcc -O2 -ffp-contract=off -shared -fPIC math_shim.c -o libmath_shim.so -lm */
#include "../../include/gie_math.h"

void ms_from_quat(const float *q, const float *t, float *out12)
{ const gie_se3 s = gie_se3_from_quat(q[0], q[1], q[2], q[3], t[0], t[1], t[2]); for (int i = 0; i < 12; i++) out12[i] = s.m[i]; }
void ms_inv(const float *in12, float *out12)
{ gie_se3 a, r; for (int i = 0; i < 12; i++) a.m[i] = in12[i]; r = gie_se3_inv(a); for (int i = 0; i < 12; i++) out12[i] = r.m[i]; }
void ms_apply(const float *m12, const float *p, float *o)
{ gie_se3 a; for (int i = 0; i < 12; i++) a.m[i] = m12[i]; gie_se3_apply(a, p[0], p[1], p[2], &o[0], &o[1], &o[2]); }
int ms_pos2coord(float p, float w) { return gie_pos2coord(p, w); }
float ms_atan2f(float y, float x) { return gie_atan2f(y, x); }
int ms_point_ok(float x, float y, float z) { return gie_point_ok(x, y, z); }
