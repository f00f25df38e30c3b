// Rotated-box geometry device functions (one lane = one box pair).
//
//  * quad_inter_f64      exact convex clip in double -- what the reference gets from shapely/GEOS
//                        (reference utils/iou_rotated_boxes_utils.py:91,119-120; utils/evaluation_utils.py:36,214)
//  * pair_term           iou_pred_vs_target_boxes for one pair (reference utils/iou_rotated_boxes_utils.py:98-142),
//                        including the float32 clip of utils/cal_intersection_rotated_boxes.py:42-96 with its exact
//                        control flow (stale polygon on a fully-rejecting edge), the 8-point
//                        hull, and the reference's PARTIAL gradient (crossing points are constants, App. A #11).
// Float32 arithmetic that decides branches is written with explicit round-to-nearest intrinsics so that
// hipcc cannot contract it into FMAs the PyTorch-CPU reference does not perform.
#pragma once
#include "common.hpp"

namespace geom {

__device__ __forceinline__ float mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub(float a, float b) { return __fsub_rn(a, b); }

// corner order: front-left, rear-left, rear-right, front-right (reference iou_rotated_boxes_utils.py:34-61)
__device__ __forceinline__ void corners(float x, float y, float w, float l, float yaw, float* cx, float* cy) {
    const float c = cosf(yaw), s = sinf(yaw);
    const float hwc = mul(w * 0.5f, c), hws = mul(w * 0.5f, s);
    const float hlc = mul(l * 0.5f, c), hls = mul(l * 0.5f, s);
    cx[0] = sub(sub(x, hwc), hls); cy[0] = add(sub(y, hws), hlc);
    cx[1] = add(sub(x, hwc), hls); cy[1] = sub(sub(y, hws), hlc);
    cx[2] = add(add(x, hwc), hls); cy[2] = sub(add(y, hws), hlc);
    cx[3] = sub(add(x, hwc), hls); cy[3] = add(add(y, hws), hlc);
}

__device__ inline double quad_inter_f64(const float* ax, const float* ay, const float* bx, const float* by) {
    double sx[16], sy[16], tx[16], ty[16], qx[4], qy[4];
    int n = 4;
    double aa = 0.0, ab = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sx[i] = (double)ax[i]; sy[i] = (double)ay[i];
        qx[i] = (double)bx[i]; qy[i] = (double)by[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = (i + 1) & 3;
        aa += sx[i] * sy[j] - sy[i] * sx[j];
        ab += qx[i] * qy[j] - qy[i] * qx[j];
    }
    if (aa == 0.0 || ab == 0.0) return 0.0;
    if (ab < 0.0) {
        double t;
        t = qx[0]; qx[0] = qx[3]; qx[3] = t; t = qy[0]; qy[0] = qy[3]; qy[3] = t;
        t = qx[1]; qx[1] = qx[2]; qx[2] = t; t = qy[1]; qy[1] = qy[2]; qy[2] = t;
    }
    for (int e = 0; e < 4 && n > 0; ++e) {
        const double cx = qx[e], cy = qy[e];
        const double ex = qx[(e + 1) & 3] - cx, ey = qy[(e + 1) & 3] - cy;
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const int j = (i + 1 == n) ? 0 : i + 1;
            const double ds = ex * (sy[i] - cy) - ey * (sx[i] - cx);
            const double dt = ex * (sy[j] - cy) - ey * (sx[j] - cx);
            if (ds >= 0.0 && m < 16) { tx[m] = sx[i]; ty[m] = sy[i]; ++m; }
            if (((ds > 0.0 && dt < 0.0) || (ds < 0.0 && dt > 0.0)) && m < 16) {
                const double u = ds / (ds - dt);
                tx[m] = sx[i] + u * (sx[j] - sx[i]);
                ty[m] = sy[i] + u * (sy[j] - sy[i]);
                ++m;
            }
        }
        n = m;
        for (int i = 0; i < n; ++i) { sx[i] = tx[i]; sy[i] = ty[i]; }
    }
    if (n < 3) return 0.0;
    double a2 = 0.0;
    for (int i = 0; i < n; ++i) {
        const int j = (i + 1 == n) ? 0 : i + 1;
        a2 += sx[i] * sy[j] - sy[i] * sx[j];
    }
    return 0.5 * fabs(a2);
}

// float32 IoU from a float64 intersection, float32 tail (see oracle/nms_ref.py iou_matrix)
__device__ __forceinline__ float iou_from_inter(double inter, float area_a, float area_b, float eps) {
    const float i32 = (float)inter;
    return i32 / add(sub(add(area_a, area_b), i32), eps);
}

struct PairOut {
    float iou, term;
    float g[6];  // d term / d (x, y, w, l, im, re) of the prediction
};

// The reference's float32 Sutherland-Hodgman with its control flow.  px/py: subject (prediction) corners,
// qx/qy: clip (target) corners.  Returns the polygon (vertex coordinates + source corner id or -1).
__device__ inline int clip_refsem(const float* px, const float* py, const float* qx, const float* qy, float* ox,
                                  float* oy, int* osrc) {
    float vx[16], vy[16], nx[16], ny[16], val[16];
    int vs[16], ns[16];
    int n = 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) { vx[i] = px[i]; vy[i] = py[i]; vs[i] = i; }
    for (int e = 0; e < 4; ++e) {
        if (n <= 2) break;
        const int e2 = (e + 1) & 3;
        const float a = sub(qy[e2], qy[e]);
        const float b = sub(qx[e], qx[e2]);
        const float c = sub(mul(qx[e2], qy[e]), mul(qy[e2], qx[e]));
        for (int i = 0; i < n; ++i) val[i] = add(add(mul(a, vx[i]), mul(b, vy[i])), c);
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const int j = (i + 1 == n) ? 0 : i + 1;
            if (val[i] <= 0.f && m < 16) { nx[m] = vx[i]; ny[m] = vy[i]; ns[m] = vs[i]; ++m; }
            if (mul(val[i], val[j]) < 0.f && m < 16) {
                const float a2 = sub(vy[j], vy[i]);
                const float b2 = sub(vx[i], vx[j]);
                const float c2 = sub(mul(vx[j], vy[i]), mul(vy[j], vx[i]));
                const float w = sub(mul(a, b2), mul(b, a2));
                nx[m] = sub(mul(b, c2), mul(c, b2)) / w;
                ny[m] = sub(mul(c, a2), mul(a, c2)) / w;
                ns[m] = -1;
                ++m;
            }
        }
        if (m == 0) break;  // reference quirk: the polygon clipped so far survives
        n = m;
        for (int i = 0; i < n; ++i) { vx[i] = nx[i]; vy[i] = ny[i]; vs[i] = ns[i]; }
    }
    for (int i = 0; i < n; ++i) { ox[i] = vx[i]; oy[i] = vy[i]; osrc[i] = vs[i]; }
    return n;
}

// shoelace area (float32) of polygon (x,y)[n]; adds coef * dA/d(vertex) into gx/gy of source corners < 4
__device__ inline float shoelace_f32(const float* x, const float* y, int n, float* sign_out) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) {
        const int j = (i + 1 == n) ? 0 : i + 1;
        s = add(s, sub(mul(x[i], y[j]), mul(y[i], x[j])));
    }
    *sign_out = s > 0.f ? 1.f : (s < 0.f ? -1.f : 0.f);
    return fabsf(s) * 0.5f;
}

// convex hull (monotone chain, float64 predicates) of 8 float32 points; returns count, indices CCW
__device__ inline int hull8(const float* x, const float* y, int* out) {
    int ord[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ord[i] = i;
    for (int i = 1; i < 8; ++i) {
        const int k = ord[i];
        int j = i - 1;
        while (j >= 0 && (x[ord[j]] > x[k] || (x[ord[j]] == x[k] && y[ord[j]] > y[k]))) { ord[j + 1] = ord[j]; --j; }
        ord[j + 1] = k;
    }
    int h[16];   // lower chain <= 8 entries, the upper adds <= 7 (16 elements stay a register vector; 18 went to scratch)
    int m = 0;
    auto cross = [&](int o, int a, int b) {
        return ((double)x[a] - (double)x[o]) * ((double)y[b] - (double)y[o]) -
               ((double)y[a] - (double)y[o]) * ((double)x[b] - (double)x[o]);
    };
    for (int i = 0; i < 8; ++i) {
        while (m >= 2 && cross(h[m - 2], h[m - 1], ord[i]) <= 0.0) --m;
        h[m++] = ord[i];
    }
    const int lower = m + 1;
    for (int i = 6; i >= 0; --i) {
        while (m >= lower && cross(h[m - 2], h[m - 1], ord[i]) <= 0.0) --m;
        h[m++] = ord[i];
    }
    --m;  // last point equals the first
    for (int i = 0; i < m; ++i) out[i] = h[i];
    return m;
}

// One (prediction, target) pair.  p/t: (x, y, w, l, im, re).  The loss variant is a template parameter so that a kernel
// which knows it at launch carries the local arrays of ONE path only (they then fit in registers: no scratch).
template <bool giou>
__device__ inline PairOut pair_term_t(const float* p, const float* t) {
    PairOut o;
    float pcx[4], pcy[4], tcx[4], tcy[4];
    const float pyaw = atan2f(p[4], p[5]);
    const float tyaw = atan2f(t[4], t[5]);
    corners(p[0], p[1], p[2], p[3], pyaw, pcx, pcy);
    corners(t[0], t[1], t[2], t[3], tyaw, tcx, tcy);
    const float parea = mul(p[2], p[3]), tarea = mul(t[2], t[3]);
    float gcx[4] = {0.f, 0.f, 0.f, 0.f}, gcy[4] = {0.f, 0.f, 0.f, 0.f};  // d term / d pred corners
    float inter;
    float dI_x[4] = {0.f, 0.f, 0.f, 0.f}, dI_y[4] = {0.f, 0.f, 0.f, 0.f};
    if (giou) {
        float vx[16], vy[16];
        int vs[16];
        const int n = clip_refsem(pcx, pcy, tcx, tcy, vx, vy, vs);
        if (n <= 2) {
            inter = 0.f;
        } else {
            float sg;
            inter = shoelace_f32(vx, vy, n, &sg);
            for (int i = 0; i < n; ++i) {
                if (vs[i] < 0) continue;
                const int nx_ = (i + 1 == n) ? 0 : i + 1, pv = (i == 0) ? n - 1 : i - 1;
                dI_x[vs[i]] = 0.5f * sg * (vy[nx_] - vy[pv]);
                dI_y[vs[i]] = 0.5f * sg * (vx[pv] - vx[nx_]);
            }
        }
    } else {
        inter = (float)quad_inter_f64(pcx, pcy, tcx, tcy);
    }
    const float uni = sub(add(parea, tarea), inter);
    const float ue = add(uni, 1e-16f);
    const float iou = inter / ue;
    float dT_dI, dT_dPa, dT_dC = 0.f;
    float term;
    if (giou) {
        float hx[8], hy[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { hx[i] = pcx[i]; hy[i] = pcy[i]; hx[4 + i] = tcx[i]; hy[4 + i] = tcy[i]; }
        int hidx[8];
        const int hn = hull8(hx, hy, hidx);
        float qx[8], qy[8];
        for (int i = 0; i < hn; ++i) { qx[i] = hx[hidx[i]]; qy[i] = hy[hidx[i]]; }
        float sg;
        const float carea = shoelace_f32(qx, qy, hn, &sg);
        const float ce = add(carea, 1e-16f);
        term = 1.f - (iou - (carea - uni) / ce);
        dT_dI = -(1.f / ue + inter / (ue * ue)) + 1.f / ce;
        dT_dPa = inter / (ue * ue) - 1.f / ce;
        dT_dC = ue / (ce * ce);
        for (int i = 0; i < hn; ++i) {
            if (hidx[i] >= 4) continue;
            const int nx_ = (i + 1 == hn) ? 0 : i + 1, pv = (i == 0) ? hn - 1 : i - 1;
            gcx[hidx[i]] += dT_dC * 0.5f * sg * (qy[nx_] - qy[pv]);
            gcy[hidx[i]] += dT_dC * 0.5f * sg * (qx[pv] - qx[nx_]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { gcx[k] += dT_dI * dI_x[k]; gcy[k] += dT_dI * dI_y[k]; }
    } else {
        term = 1.f - iou;
        dT_dI = 0.f;
        dT_dPa = inter / (ue * ue);
    }
    (void)dT_dI;
    // chain corners -> (x, y, w, l, yaw)
    const float c = cosf(pyaw), s = sinf(pyaw);
    const float sa[4] = {-1.f, -1.f, 1.f, 1.f}, sb[4] = {-1.f, 1.f, 1.f, -1.f};
    float gx = 0.f, gy = 0.f, gw = 0.f, gl = 0.f, gyaw = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        gx += gcx[k];
        gy += gcy[k];
        gw += gcx[k] * (sa[k] * c * 0.5f) + gcy[k] * (sa[k] * s * 0.5f);
        gl += gcx[k] * (sb[k] * s * 0.5f) + gcy[k] * (-sb[k] * c * 0.5f);
        gyaw += gcx[k] * (-sa[k] * p[2] * 0.5f * s + sb[k] * p[3] * 0.5f * c) +
                gcy[k] * (sa[k] * p[2] * 0.5f * c + sb[k] * p[3] * 0.5f * s);
    }
    gw += dT_dPa * p[3];
    gl += dT_dPa * p[2];
    const float r2 = p[4] * p[4] + p[5] * p[5];
    o.iou = iou;
    o.term = term;
    o.g[0] = gx; o.g[1] = gy; o.g[2] = gw; o.g[3] = gl;
    o.g[4] = gyaw * (p[5] / r2);
    o.g[5] = gyaw * (-p[4] / r2);
    return o;
}

__device__ inline PairOut pair_term(const float* p, const float* t, bool giou) {
    return giou ? pair_term_t<true>(p, t) : pair_term_t<false>(p, t);
}

}  // namespace geom
