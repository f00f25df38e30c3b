// Rotated-box geometry device functions (one lane = one box pair; the calling kernel provides a geom::Pool, see below).
//
//  * quad_inter_f64      exact convex clip in double -- what the reference gets from shapely/GEOS
//                        (reference utils/iou_rotated_boxes_utils.py:91,119-120; utils/evaluation_utils.py:36,214)
//  * pair_term           iou_pred_vs_target_boxes for one pair (reference utils/iou_rotated_boxes_utils.py:98-142),
//                        including the float32 clip of utils/cal_intersection_rotated_boxes.py:42-96 with its exact
//                        control flow (stale polygon on a fully-rejecting edge), the 8-point
//                        hull, and the reference's PARTIAL gradient (crossing points are constants, App. A #11).
// Float32 arithmetic that decides branches goes through mul / add / sub below so that hipcc cannot contract it into FMAs
// the PyTorch-CPU reference does not perform.
#pragma once
#include "common.hpp"

namespace geom {

// (hipcc contracts a float multiply feeding an add into one FMA even through __fmul_rn / __fadd_rn -- see bev.hip -- so the
// product is pinned in a register by an empty asm: every float32 operation of the clip is rounded on its own, as in the
// reference's tensor arithmetic; degenerate inputs such as identical boxes depend on it)
__device__ __forceinline__ float mul(float a, float b) {
    float m = a * b;
    asm volatile("" : "+v"(m));
    return m;
}
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

// ---- lane-private work arrays in LDS --------------------------------------------------------------------------------
// The polygon routines below index small per-lane arrays with data-dependent indices (vertex lists that grow and shrink).
// As register arrays hipcc serves such an index through the VGPR-index mode (s_set_gpr_idx_on ... s_set_gpr_idx_off), as
// stack arrays through scratch memory; here every dynamically indexed array lives in LDS, element i of lane l at word
// (slot * 16 + i) * 64 + l (conflict-free, a data-dependent index is plain address arithmetic), and what stays in registers is
// only ever indexed by unrolled loop counters: no private segment, no index mode.  32 KB per 64-thread block; doubles pair
// two slots.  (Rounds 2-3 moved the arrays here while hunting results that changed in lanes 48-63 of a wave whenever another
// kernel ran beside these ones.  The storage was not the cause: the SLP-packed float32 arithmetic was -- see build.py
// (-fno-slp-vectorize for this code) and profiles/r05_head_race.txt.)
constexpr int POOL_LANES = 64, POOL_SLOTS = 8, POOL_LEN = 16;
constexpr int POOL_BYTES = POOL_SLOTS * POOL_LEN * POOL_LANES * 4;
struct Pool {
    unsigned char* base;
    int lane;
    __device__ __forceinline__ float& f(int slot, int i) const { return reinterpret_cast<float*>(base)[(slot * POOL_LEN + i) * POOL_LANES + lane]; }
    __device__ __forceinline__ int& n(int slot, int i) const { return reinterpret_cast<int*>(base)[(slot * POOL_LEN + i) * POOL_LANES + lane]; }
    __device__ __forceinline__ double& d(int slot, int i) const { return reinterpret_cast<double*>(base)[(slot * POOL_LEN + i) * POOL_LANES + lane]; }  // slot < 4
};
// declares the calling kernel's pool (blocks of at most 64 threads)
#define CY_GEOM_POOL(name)                                                                  \
    __shared__ __attribute__((aligned(16))) unsigned char name##_mem[geom::POOL_BYTES];     \
    const geom::Pool name = {name##_mem, (int)(threadIdx.x & 63)}

__device__ inline double quad_inter_f64(const Pool& P, const float* ax, const float* ay, const float* bx, const float* by) {
    // subject polygon in the double slots (0, 1), clipped into (2, 3) and back, edge by edge
    double qx[4], qy[4];
    int n = 4;
    double aa = 0.0, ab = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        P.d(0, i) = (double)ax[i]; P.d(1, i) = (double)ay[i];
        qx[i] = (double)bx[i]; qy[i] = (double)by[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = (i + 1) & 3;
        aa += (double)ax[i] * (double)ay[j] - (double)ay[i] * (double)ax[j];
        ab += qx[i] * qy[j] - qy[i] * qx[j];
    }
    if (aa == 0.0 || ab == 0.0) return 0.0;
    if (ab < 0.0) {
        double t;
        t = qx[0]; qx[0] = qx[3]; qx[3] = t; t = qy[0]; qy[0] = qy[3]; qy[3] = t;
        t = qx[1]; qx[1] = qx[2]; qx[2] = t; t = qy[1]; qy[1] = qy[2]; qy[2] = t;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (n <= 0) continue;
        const int src = (e & 1) ? 2 : 0, dst = 2 - src;
        const double cx = qx[e], cy = qy[e];
        const double ex = qx[(e + 1) & 3] - cx, ey = qy[(e + 1) & 3] - cy;
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const int j = (i + 1 == n) ? 0 : i + 1;
            const double sxi = P.d(src, i), syi = P.d(src + 1, i), sxj = P.d(src, j), syj = P.d(src + 1, j);
            const double ds = ex * (syi - cy) - ey * (sxi - cx);
            const double dt = ex * (syj - cy) - ey * (sxj - cx);
            if (ds >= 0.0 && m < 16) { P.d(dst, m) = sxi; P.d(dst + 1, m) = syi; ++m; }
            if (((ds > 0.0 && dt < 0.0) || (ds < 0.0 && dt > 0.0)) && m < 16) {
                const double u = ds / (ds - dt);
                P.d(dst, m) = sxi + u * (sxj - sxi);
                P.d(dst + 1, m) = syi + u * (syj - syi);
                ++m;
            }
        }
        n = m;
    }
    if (n < 3) return 0.0;
    // four edges processed: the polygon is back in slots (0, 1)
    double a2 = 0.0;
    for (int i = 0; i < n; ++i) {
        const int j = (i + 1 == n) ? 0 : i + 1;
        a2 += P.d(0, i) * P.d(1, j) - P.d(1, i) * P.d(0, j);
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

// The reference's float32 Sutherland-Hodgman with its control flow.  px/py: subject (prediction) corners, qx/qy: clip (target)
// corners.  Leaves the polygon in the pool -- x in slot *cur, y in *cur + 1, source corner id (or -1) in *cur + 2 -- and
// returns its vertex count.  Slots 0-2 and 3-5 alternate as source and destination, slot 6 holds the edge function values.
__device__ inline int clip_refsem(const Pool& P, const float* px, const float* py, const float* qx, const float* qy, int* cur_out) {
    int n = 4, cur = 0;
    bool done = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) { P.f(0, i) = px[i]; P.f(1, i) = py[i]; P.n(2, i) = i; }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (done || n <= 2) continue;
        const int e2 = (e + 1) & 3, dst = 3 - cur;
        const float a = sub(qy[e2], qy[e]);
        const float b = sub(qx[e], qx[e2]);
        const float c = sub(mul(qx[e2], qy[e]), mul(qy[e2], qx[e]));
        for (int i = 0; i < n; ++i) P.f(6, i) = add(add(mul(a, P.f(cur, i)), mul(b, P.f(cur + 1, i))), c);
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const int j = (i + 1 == n) ? 0 : i + 1;
            const float vi = P.f(6, i), vj = P.f(6, j);
            const float xi = P.f(cur, i), yi = P.f(cur + 1, i);
            if (vi <= 0.f && m < 16) { P.f(dst, m) = xi; P.f(dst + 1, m) = yi; P.n(dst + 2, m) = P.n(cur + 2, i); ++m; }
            if (mul(vi, vj) < 0.f && m < 16) {
                const float xj = P.f(cur, j), yj = P.f(cur + 1, j);
                const float a2 = sub(yj, yi);
                const float b2 = sub(xi, xj);
                const float c2 = sub(mul(xj, yi), mul(yj, xi));
                const float w = sub(mul(a, b2), mul(b, a2));
                P.f(dst, m) = sub(mul(b, c2), mul(c, b2)) / w;
                P.f(dst + 1, m) = sub(mul(c, a2), mul(a, c2)) / w;
                P.n(dst + 2, m) = -1;
                ++m;
            }
        }
        if (m == 0) { done = true; continue; }  // reference quirk: the polygon clipped so far survives
        n = m;
        cur = dst;
    }
    *cur_out = cur;
    return n;
}

// shoelace area (float32) of the polygon in pool slots (sx, sy)[n]
__device__ inline float shoelace_f32(const Pool& P, int sx, int sy, int n, float* sign_out) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) {
        const int j = (i + 1 == n) ? 0 : i + 1;
        s = add(s, sub(mul(P.f(sx, i), P.f(sy, j)), mul(P.f(sy, i), P.f(sx, j))));
    }
    *sign_out = s > 0.f ? 1.f : (s < 0.f ? -1.f : 0.f);
    return fabsf(s) * 0.5f;
}

// convex hull (monotone chain, float64 predicates) of the 8 float32 points in pool slots (0, 1); order in slot 2, chain in
// slot 3 (lower chain <= 8 entries, the upper adds <= 7), result -- indices CCW -- in slot 4; returns the count
__device__ inline int hull8(const Pool& P) {
#pragma unroll
    for (int i = 0; i < 8; ++i) P.n(2, i) = i;
    for (int i = 1; i < 8; ++i) {
        const int k = P.n(2, i);
        const float xk = P.f(0, k), yk = P.f(1, k);
        int j = i - 1;
        while (j >= 0) {
            const int oj = P.n(2, j);
            const float xo = P.f(0, oj);
            if (!(xo > xk || (xo == xk && P.f(1, oj) > yk))) break;
            P.n(2, j + 1) = oj;
            --j;
        }
        P.n(2, j + 1) = k;
    }
    int m = 0;
    auto cross = [&](int o, int a, int b) {
        return ((double)P.f(0, a) - (double)P.f(0, o)) * ((double)P.f(1, b) - (double)P.f(1, o)) -
               ((double)P.f(1, a) - (double)P.f(1, o)) * ((double)P.f(0, b) - (double)P.f(0, o));
    };
    for (int i = 0; i < 8; ++i) {
        const int oi = P.n(2, i);
        while (m >= 2 && cross(P.n(3, m - 2), P.n(3, m - 1), oi) <= 0.0) --m;
        P.n(3, m++) = oi;
    }
    const int lower = m + 1;
    for (int i = 6; i >= 0; --i) {
        const int oi = P.n(2, i);
        while (m >= lower && cross(P.n(3, m - 2), P.n(3, m - 1), oi) <= 0.0) --m;
        P.n(3, m++) = oi;
    }
    --m;  // last point equals the first
    for (int i = 0; i < m; ++i) P.n(4, i) = P.n(3, i);
    return m;
}

// One (prediction, target) pair.  p/t: (x, y, w, l, im, re).  The term is assembled from two independent halves -- the
// INTERSECTION (the reference's float32 clip, its area and d area / d prediction corners) and the 8-point HULL (its area and
// the hull-edge differences at the prediction's corners) -- so that a kernel may give them to two lanes of a wave
// (yolo_head.hip::pairs_kernel) and join them with shuffles; pair_term_t runs both in one lane.  Every float32 operation keeps
// the order it had in the single-lane form: the two ways produce identical bits.
struct InterPart {
    float inter;
    float dIx[4], dIy[4];   // d inter / d prediction corner (crossing points are constants: reference App. A #11)
};
struct HullPart {
    float carea, sg;        // hull area, orientation sign
    float dy[4], dx[4];     // (qy[next] - qy[prev]), (qx[prev] - qx[next]) at prediction corner k when it is a hull vertex
    int on;                 // bit k: prediction corner k is a hull vertex
};

template <bool giou>
__device__ inline InterPart inter_part(const Pool& P, const float* pcx, const float* pcy, const float* tcx, const float* tcy) {
    InterPart r;
#pragma unroll
    for (int k = 0; k < 4; ++k) { r.dIx[k] = 0.f; r.dIy[k] = 0.f; }
    if (giou) {
        int cur;
        const int n = clip_refsem(P, pcx, pcy, tcx, tcy, &cur);
        if (n <= 2) {
            r.inter = 0.f;
        } else {
            float sg;
            r.inter = shoelace_f32(P, cur, cur + 1, n, &sg);
            for (int i = 0; i < n; ++i) {
                const int src = P.n(cur + 2, i);
                if (src < 0) continue;
                const int nx_ = (i + 1 == n) ? 0 : i + 1, pv = (i == 0) ? n - 1 : i - 1;
                const float gx = 0.5f * sg * (P.f(cur + 1, nx_) - P.f(cur + 1, pv)), gy = 0.5f * sg * (P.f(cur, pv) - P.f(cur, nx_));
#pragma unroll
                for (int k = 0; k < 4; ++k)      // (selects instead of a dynamic index into registers)
                    if (src == k) { r.dIx[k] = gx; r.dIy[k] = gy; }
            }
        }
    } else {
        r.inter = (float)quad_inter_f64(P, pcx, pcy, tcx, tcy);
    }
    return r;
}

__device__ inline HullPart hull_part(const Pool& P, const float* pcx, const float* pcy, const float* tcx, const float* tcy) {
    HullPart h;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        P.f(0, i) = pcx[i]; P.f(1, i) = pcy[i]; P.f(0, 4 + i) = tcx[i]; P.f(1, 4 + i) = tcy[i];
        h.dy[i] = 0.f; h.dx[i] = 0.f;
    }
    const int hn = hull8(P);
    for (int i = 0; i < hn; ++i) { const int id = P.n(4, i); P.f(5, i) = P.f(0, id); P.f(6, i) = P.f(1, id); }
    h.carea = shoelace_f32(P, 5, 6, hn, &h.sg);
    h.on = 0;
    for (int i = 0; i < hn; ++i) {
        const int id = P.n(4, i);
        if (id >= 4) continue;
        const int nx_ = (i + 1 == hn) ? 0 : i + 1, pv = (i == 0) ? hn - 1 : i - 1;
        const float ddy = P.f(6, nx_) - P.f(6, pv), ddx = P.f(5, pv) - P.f(5, nx_);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (id == k) { h.dy[k] = ddy; h.dx[k] = ddx; h.on |= 1 << k; }
    }
    return h;
}

template <bool giou>
__device__ inline PairOut pair_finish(const float* p, const float* t, float pyaw, const InterPart& ip, const HullPart& hp) {
    PairOut o;
    const float parea = mul(p[2], p[3]), tarea = mul(t[2], t[3]);
    float gcx[4] = {0.f, 0.f, 0.f, 0.f}, gcy[4] = {0.f, 0.f, 0.f, 0.f};  // d term / d pred corners
    const float inter = ip.inter;
    const float uni = sub(add(parea, tarea), inter);
    const float ue = add(uni, 1e-16f);
    const float iou = inter / ue;
    float dT_dI, dT_dPa, dT_dC = 0.f;
    float term;
    if (giou) {
        const float carea = hp.carea, sg = hp.sg;
        const float ce = add(carea, 1e-16f);
        term = 1.f - (iou - (carea - uni) / ce);
        dT_dI = -(1.f / ue + inter / (ue * ue)) + 1.f / ce;
        dT_dPa = inter / (ue * ue) - 1.f / ce;
        dT_dC = ue / (ce * ce);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!((hp.on >> k) & 1)) continue;
            gcx[k] += dT_dC * 0.5f * sg * hp.dy[k];
            gcy[k] += dT_dC * 0.5f * sg * hp.dx[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { gcx[k] += dT_dI * ip.dIx[k]; gcy[k] += dT_dI * ip.dIy[k]; }
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

// the single-lane form.  The loss variant is a template parameter so that a kernel which knows it at launch carries the local
// arrays of ONE path only (they then fit in registers: no scratch).
template <bool giou>
__device__ inline PairOut pair_term_t(const Pool& P, const float* p, const float* t) {
    float pcx[4], pcy[4], tcx[4], tcy[4];
    const float pyaw = atan2f(p[4], p[5]);
    const float tyaw = atan2f(t[4], t[5]);
    corners(p[0], p[1], p[2], p[3], pyaw, pcx, pcy);
    corners(t[0], t[1], t[2], t[3], tyaw, tcx, tcy);
    const InterPart ip = inter_part<giou>(P, pcx, pcy, tcx, tcy);
    HullPart hp;
    hp.carea = 0.f; hp.sg = 0.f; hp.on = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { hp.dy[k] = 0.f; hp.dx[k] = 0.f; }
    if (giou) hp = hull_part(P, pcx, pcy, tcx, tcy);
    return pair_finish<giou>(p, t, pyaw, ip, hp);
}

__device__ inline PairOut pair_term(const Pool& P, const float* p, const float* t, bool giou) {
    return giou ? pair_term_t<true>(P, p, t) : pair_term_t<false>(P, p, t);
}

}  // namespace geom
