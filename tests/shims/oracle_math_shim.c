/* TEST-ONLY: exports oracle/oracle_math.h (the ORACLE's own statement of the fp32 geometry) behind the interface of math_shim.c,
 * so that tests/test_independent_checks.py runs every check of the product's include/gie_math.h on it as well and holds the two
 * against each other bit for bit.
 *   gcc -O2 -ffp-contract=off -shared -fPIC oracle_math_shim.c -o liboracle_math_shim.so -lm */
#include "../../oracle/oracle_math.h"

static om_pose from12(const float *m) { om_pose p; for (int i = 0; i < 3; i++) { for (int k = 0; k < 3; k++) p.r[i][k] = m[4 * i + k]; p.t[i] = m[4 * i + 3]; } return p; }
static void to12(const om_pose p, float *m) { for (int i = 0; i < 3; i++) { for (int k = 0; k < 3; k++) m[4 * i + k] = p.r[i][k]; m[4 * i + 3] = p.t[i]; } }

void ms_from_quat(const float *q, const float *t, float *out12) { to12(om_from_quat(q, t), out12); }
void ms_inv(const float *in12, float *out12) { to12(om_inverse(from12(in12)), out12); }
void ms_apply(const float *m12, const float *p, float *o) { om_transform(from12(m12), p, o); }
int ms_pos2coord(float p, float w) { return om_voxel_of(p, w); }
float ms_atan2f(float y, float x) { return om_atan2(y, x); }
int ms_point_ok(float x, float y, float z) { const float g[3] = { x, y, z }; return om_point_usable(g); }
