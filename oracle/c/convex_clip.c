/*
 * ORACLE (test infrastructure, not product code).
 *
 * Float64 area of the intersection of two convex quadrilaterals given as float32 corner lists.
 * This stands where the reference calls shapely/GEOS:
 *   Polygon(corners).buffer(0).intersection(other).area
 *     reference src/utils/iou_rotated_boxes_utils.py:24-31,91,119-120
 *     reference src/utils/evaluation_utils.py:15-21,36,214
 * shapely (GEOS) is an un-vendored, unpinned dependency of the reference (not in requirements.txt),
 * so its published behaviour is restated: for two valid convex polygons the intersection is the
 * convex polygon obtained by clipping one against the half-planes of the other; the area is taken
 * in IEEE double.  Orientation-agnostic (a clockwise ring is reversed first); a ring of zero area
 * has an empty interior (what buffer(0) returns) and intersects nothing.
 *
 * PARITY UNPINNED against GEOS itself (absent from the container); pinned only against the
 * analytic known answers in tests/golden.
 */
#include <math.h>
#include <stddef.h>

#define MAXV 16

static double ring_area2(const double* p, int n) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        int j = (i + 1 == n) ? 0 : i + 1;
        s += p[2 * i] * p[2 * j + 1] - p[2 * i + 1] * p[2 * j];
    }
    return s;
}

double cy_oracle_quad_inter_area(const float* qa, const float* qb) {
    double subj[2 * MAXV], tmp[2 * MAXV], clip[8];
    int n = 4;
    for (int i = 0; i < 8; ++i) subj[i] = (double)qa[i];
    for (int i = 0; i < 8; ++i) clip[i] = (double)qb[i];
    double ab = ring_area2(clip, 4);
    if (ring_area2(subj, 4) == 0.0 || ab == 0.0) return 0.0;
    if (ab < 0.0) { /* make the clip ring counter-clockwise */
        for (int i = 0; i < 2; ++i) {
            int j = 3 - i;
            double tx = clip[2 * i], ty = clip[2 * i + 1];
            clip[2 * i] = clip[2 * j]; clip[2 * i + 1] = clip[2 * j + 1];
            clip[2 * j] = tx; clip[2 * j + 1] = ty;
        }
    }
    for (int e = 0; e < 4 && n > 0; ++e) {
        const double cx = clip[2 * e], cy = clip[2 * e + 1];
        const int e2 = (e + 1) & 3;
        const double ex = clip[2 * e2] - cx, ey = clip[2 * e2 + 1] - cy;
        int m = 0;
        for (int i = 0; i < n; ++i) {
            int j = (i + 1 == n) ? 0 : i + 1;
            const double sx = subj[2 * i], sy = subj[2 * i + 1];
            const double tx = subj[2 * j], ty = subj[2 * j + 1];
            const double ds = ex * (sy - cy) - ey * (sx - cx);
            const double dt = ex * (ty - cy) - ey * (tx - cx);
            if (ds >= 0.0) { tmp[2 * m] = sx; tmp[2 * m + 1] = sy; ++m; }
            if ((ds > 0.0 && dt < 0.0) || (ds < 0.0 && dt > 0.0)) {
                const double u = ds / (ds - dt);
                tmp[2 * m] = sx + u * (tx - sx); tmp[2 * m + 1] = sy + u * (ty - sy); ++m;
            }
        }
        n = m;
        for (int i = 0; i < 2 * n; ++i) subj[i] = tmp[i];
    }
    if (n < 3) return 0.0;
    return 0.5 * fabs(ring_area2(subj, n));
}

void cy_oracle_inter_pairs(const float* a, const float* b, long n, double* out) {
    for (long i = 0; i < n; ++i) out[i] = cy_oracle_quad_inter_area(a + 8 * i, b + 8 * i);
}

void cy_oracle_inter_matrix(const float* a, long na, const float* b, long nb, double* out) {
    for (long i = 0; i < na; ++i)
        for (long j = 0; j < nb; ++j) out[i * nb + j] = cy_oracle_quad_inter_area(a + 8 * i, b + 8 * j);
}
