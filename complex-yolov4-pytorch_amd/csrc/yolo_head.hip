// YOLO head on device: decode, fused target assignment + loss + metrics + d(loss)/d(logits).
// Reference: models/yolo_layer.py (decode :144-189, build_targets :69-142, loss/metrics :199-251).
// Logits arrive NHWC fp32 straight from the head conv: [B][G][G][A*(7+C)], so one cell's A*(7+C)
// values are contiguous.  Targets are few (nT ~ 100): assignment is sparse (one lane per target,
// atomics for the per-cell ownership maps); only the objectness BCE and the metric counters are dense.
// All sums accumulate in double atomics, so the value does not depend on a float32 reduction tree.
#include "common.hpp"
#include "geometry.hpp"

namespace {

constexpr int MAXA = 8;
struct Anchors {
    float w[MAXA], h[MAXA], im[MAXA], re[MAXA];  // w,h already divided by the stride (grid units)
};

enum Acc {
    A_SX, A_SY, A_SW, A_SH, A_SIM, A_SRE, A_UNIT, A_BCE_OBJ, A_BCE_NOOBJ, A_CLS, A_GIOU, A_IOU, A_CLSACC,
    A_CONF_OBJ, A_CONF_NOOBJ, A_CONF50, A_DET50, A_DET75, A_COUNT
};
enum Cnt { C_NOBJ, C_NCLEARED, C_ERR, C_COUNT };

struct Work {
    int* owner;     // [cells]  max(target index + 1) that claimed the cell
    int* flags;     // [cells]  bit0: noobj cleared; bit (8+c): class c present
    double* acc;    // [A_COUNT]
    int* cnt;       // [C_COUNT]
    int* ti;        // [nT][4]  b, best anchor, gj, gi  (b = -1: rejected)
    float* tf;      // [nT][8]  iou, term, g[6]
    float* part;    // [2][nT][12]  the two halves of a pair term: intersection (inter, dIx[4], dIy[4]) / hull (carea, sg, dy[4], dx[4], on)
};

// Everything that differs between the heads of one model, for the three-heads-in-one-launch forms (cy_yolo_loss_multi): passed by
// value, head = blockIdx.y (a switch over three kernel arguments, so a lane-varying anchor index stays a load from the kernel
// argument segment exactly as in the single-head kernels).
struct HeadP {
    const float* logits;
    float* dlogits;
    float* metrics;
    int G, row_offset;
    float stride;
    long cells;
    Anchors an_dec, an;      // decode anchors (w, h in grid units), loss anchors (w, h in grid units, im, re)
    Work w;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ void decode_body(const float* __restrict__ logits, int B, int G, int A, int C, const Anchors& an, float stride,
                                            float* __restrict__ out, int rows_total, int row_offset) {
    const int NCH = 7 + C;
    const long total = (long)B * G * G * A;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int a = (int)(i % A);
        const long cell = i / A;
        const int gx = (int)(cell % G), gy = (int)((cell / G) % G), b = (int)(cell / ((long)G * G));
        const float* t = logits + cell * (A * NCH) + a * NCH;
        float* o = out + ((long)b * rows_total + row_offset + ((long)a * G + gy) * G + gx) * NCH;
        o[0] = (sigmoidf_(t[0]) + (float)gx) * stride;
        o[1] = (sigmoidf_(t[1]) + (float)gy) * stride;
        o[2] = fminf(expf(t[2]), 1e3f) * an.w[a] * stride;
        o[3] = fminf(expf(t[3]), 1e3f) * an.h[a] * stride;
        o[4] = t[4];
        o[5] = t[5];
        for (int c = 6; c < NCH; ++c) o[c] = sigmoidf_(t[c]);
    }
}
__global__ void decode_kernel(const float* __restrict__ logits, int B, int G, int A, int C, Anchors an, float stride,
                              float* __restrict__ out, int rows_total, int row_offset) {
    decode_body(logits, B, G, A, C, an, stride, out, rows_total, row_offset);
}
#define CY_HEAD_SWITCH(call)                                                       \
    if (blockIdx.y == 0) { const HeadP& h = h0; call; }                           \
    else if (blockIdx.y == 1) { const HeadP& h = h1; call; }                      \
    else { const HeadP& h = h2; call; }
__global__ void decode3_kernel(HeadP h0, HeadP h1, HeadP h2, int B, int A, int C, float* __restrict__ out, int rows_total) {
    CY_HEAD_SWITCH(decode_body(h.logits, B, h.G, A, C, h.an_dec, h.stride, out, rows_total, h.row_offset))
}

// Wave-cooperative target assignment (reference yolo_layer.py:69-142 with iou_rotated_boxes_utils.py:64-96 per anchor): LPT = 4
// lanes per target, lane `sub` clips the target against anchors sub, sub + 4, ... (the float64 convex clip, the long pole of
// the head: three of them in series per target were 57-62 us of pure latency per launch), then the best anchor is an argmax
// over the 4 lanes by __shfl_xor with the reference's tie rule (the FIRST maximum in anchor order).  Every lane raises the
// "not no-object" flags of its own anchors; lane 0 of the target records the assignment and claims the cell.
// (64-thread blocks, stated to the compiler: with the register budget of a single wave per SIMD the polygon arrays of the
// per-target kernels need no scratch memory.)
constexpr int LPT = 4;

// Number of live target rows of a launch.  ncap: the count the launch was SIZED for (grid, workspace layout); nT_dev: optional
// device word holding the batch's real count (<= ncap).  With it one recorded launch list (cy_run_plan) serves every batch whose
// count falls into the same bucket -- KITTI batches differ in their number of boxes almost every step -- and rows
// [count, ncap) of the target buffer are never looked at.
__device__ __forceinline__ int live_rows(int ncap, const int* __restrict__ nT_dev) {
    if (!nT_dev) return ncap;
    const int n = *nT_dev;
    return n < 0 ? 0 : (n > ncap ? ncap : n);
}
__device__ __forceinline__ void assign_body(const geom::Pool& P, const float* __restrict__ targets, int ncap, const int* __restrict__ nT_dev,
                                            int B, int G, int A, const Anchors& an, float ignore_thresh, const Work& w) {
    const int nT = live_rows(ncap, nT_dev);
    const int lane = threadIdx.x, sub = lane & (LPT - 1);
    const int k = blockIdx.x * (64 / LPT) + (lane >> 2);
    const bool live = k < nT;
    const float* t = targets + (long)(live ? k : 0) * 8;
    const int b = (int)t[0], label = (int)t[1];
    const float gf = (float)G;
    const float x = t[2] * gf, y = t[3] * gf, tw = t[4] * gf, tl = t[5] * gf;
    const int gi = (int)x, gj = (int)y;
    const bool bad = b < 0 || b >= B || gi < 0 || gi >= G || gj < 0 || gj >= G || label < 0 || label > 22;
    float tcx[4], tcy[4], acx[4], acy[4];
    geom::corners(100.f, 100.f, tw, tl, atan2f(t[6], t[7]), tcx, tcy);
    const float tarea = tw * tl;
    float best_iou = -1.f;
    int best = 0;
    float ious[MAXA / LPT];
#pragma unroll
    for (int i = 0; i < MAXA / LPT; ++i) {
        const int a = sub + LPT * i;
        ious[i] = -2.f;
        if (a >= A) continue;          // (wave-uniform per i only when A is a multiple of LPT; the clip below is per lane anyway)
        geom::corners(100.f, 100.f, an.w[a], an.h[a], atan2f(an.im[a], an.re[a]), acx, acy);
        const double inter = geom::quad_inter_f64(P, acx, acy, tcx, tcy);
        ious[i] = geom::iou_from_inter(inter, an.w[a] * an.h[a], tarea, 1e-16f);
        if (ious[i] > best_iou) { best_iou = ious[i]; best = a; }
    }
    // argmax over the target's lanes: larger IoU wins, equal IoUs keep the lower anchor (= the reference's first maximum)
#pragma unroll
    for (int m = 1; m < LPT; m <<= 1) {
        const float oi = __shfl_xor(best_iou, m);
        const int oa = __shfl_xor(best, m);
        if (oi > best_iou || (oi == best_iou && oa < best)) { best_iou = oi; best = oa; }
    }
    if (!live) return;
    if (bad) {
        if (sub == 0) {
            w.ti[k * 4] = -1;
            atomicAdd(&w.cnt[C_ERR], 1);
        }
        return;
    }
    if (sub == 0) { w.ti[k * 4 + 0] = b; w.ti[k * 4 + 1] = best; w.ti[k * 4 + 2] = gj; w.ti[k * 4 + 3] = gi; }
#pragma unroll
    for (int i = 0; i < MAXA / LPT; ++i) {
        const int a = sub + LPT * i;
        if (a >= A) continue;
        if (a != best && !(ious[i] > ignore_thresh)) continue;
        const int cell = ((b * A + a) * G + gj) * G + gi;
        const int old = atomicOr(&w.flags[cell], 1);
        if (!(old & 1)) atomicAdd(&w.cnt[C_NCLEARED], 1);
    }
    if (sub == 0) {
        const int cell = ((b * A + best) * G + gj) * G + gi;
        atomicOr(&w.flags[cell], 1 << (8 + label));
        const int old = atomicMax(&w.owner[cell], k + 1);
        if (old == 0) atomicAdd(&w.cnt[C_NOBJ], 1);
    }
}

__global__ void __launch_bounds__(64) assign_kernel(const float* __restrict__ targets, int nT, int B, int G, int A, Anchors an,
                                                    float ignore_thresh, Work w) {
    CY_GEOM_POOL(P);
    assign_body(P, targets, nT, nullptr, B, G, A, an, ignore_thresh, w);
}
__global__ void __launch_bounds__(64) assign3_kernel(HeadP h0, HeadP h1, HeadP h2, const float* __restrict__ targets, int ncap,
                                                     const int* __restrict__ nT_dev, int B, int A, float ignore_thresh) {
    CY_GEOM_POOL(P);
    CY_HEAD_SWITCH(assign_body(P, targets, ncap, nT_dev, B, h.G, A, h.an, ignore_thresh, h.w))
}

__device__ __forceinline__ void decode_box(const float* t, int gi, int gj, float aw, float ah, float* box) {
    box[0] = sigmoidf_(t[0]) + (float)gi;
    box[1] = sigmoidf_(t[1]) + (float)gj;
    box[2] = fminf(expf(t[2]), 1e3f) * aw;
    box[3] = fminf(expf(t[3]), 1e3f) * ah;
    box[4] = t[4];
    box[5] = t[5];
}

// Prediction-vs-target IoU / GIoU term and its gradient (reference iou_rotated_boxes_utils.py:98-142).  The term has two
// independent halves (geometry.hpp): the reference's float32 polygon clip with the intersection area, and the 8-point convex
// hull (GIoU's enclosing area).  They run as two halves of ONE grid -- blocks [0, nb) the clips, blocks [nb, 2 nb) the hulls,
// one lane per target each, every wave executing a single code path (two lanes of a wave would run the halves one after the
// other: divergence serialises them) -- and leave their results in the workspace; pairs_finish_kernel joins them.
template <bool GIOU>
__device__ __forceinline__ void pairs_body(const geom::Pool& P, const float* __restrict__ logits, const float* __restrict__ targets, int ncap,
                                           const int* __restrict__ nT_dev, int G, int A, int C, const Anchors& an, const Work& w, int nb) {
    const int part = (int)blockIdx.x >= nb ? 1 : 0;
    const int k = ((int)blockIdx.x - part * nb) * 64 + (int)threadIdx.x;
    if (k >= live_rows(ncap, nT_dev)) return;
    const int b = w.ti[k * 4];
    if (b < 0) return;
    const int a = w.ti[k * 4 + 1], gj = w.ti[k * 4 + 2], gi = w.ti[k * 4 + 3];
    const int NCH = 7 + C;
    const float* lg = logits + ((long)(b * G + gj) * G + gi) * (A * NCH) + a * NCH;
    float pb[6], tb[6];
    decode_box(lg, gi, gj, an.w[a], an.h[a], pb);
    const float* t = targets + (long)k * 8;
    const float gf = (float)G;
    tb[0] = t[2] * gf; tb[1] = t[3] * gf; tb[2] = t[4] * gf; tb[3] = t[5] * gf; tb[4] = t[6]; tb[5] = t[7];
    float pcx[4], pcy[4], tcx[4], tcy[4];
    geom::corners(pb[0], pb[1], pb[2], pb[3], atan2f(pb[4], pb[5]), pcx, pcy);
    geom::corners(tb[0], tb[1], tb[2], tb[3], atan2f(tb[4], tb[5]), tcx, tcy);
    float* out = w.part + ((long)part * ncap + k) * 12;      // (the two halves are ncap rows apart: the layout follows the capacity)
    if (part == 0) {
        const geom::InterPart ip = geom::inter_part<GIOU>(P, pcx, pcy, tcx, tcy);
        out[0] = ip.inter;
#pragma unroll
        for (int i = 0; i < 4; ++i) { out[1 + i] = ip.dIx[i]; out[5 + i] = ip.dIy[i]; }
    } else {
        const geom::HullPart hp = geom::hull_part(P, pcx, pcy, tcx, tcy);
        out[0] = hp.carea; out[1] = hp.sg;
#pragma unroll
        for (int i = 0; i < 4; ++i) { out[2 + i] = hp.dy[i]; out[6 + i] = hp.dx[i]; }
        out[10] = __int_as_float(hp.on);
    }
}

template <bool GIOU>
__global__ void __launch_bounds__(64) pairs_kernel(const float* __restrict__ logits, const float* __restrict__ targets, int nT,
                                                   int G, int A, int C, Anchors an, Work w, int nb) {
    CY_GEOM_POOL(P);
    pairs_body<GIOU>(P, logits, targets, nT, nullptr, G, A, C, an, w, nb);
}
template <bool GIOU>
__global__ void __launch_bounds__(64) pairs3_kernel(HeadP h0, HeadP h1, HeadP h2, const float* __restrict__ targets, int ncap,
                                                    const int* __restrict__ nT_dev, int A, int C, int nb) {
    CY_GEOM_POOL(P);
    CY_HEAD_SWITCH(pairs_body<GIOU>(P, h.logits, targets, ncap, nT_dev, h.G, A, C, h.an, h.w, nb))
}

template <bool GIOU>
__device__ __forceinline__ void pairs_finish_body(const float* __restrict__ logits, const float* __restrict__ targets, int ncap,
                                                  const int* __restrict__ nT_dev, int G, int A, int C, const Anchors& an, const Work& w) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= live_rows(ncap, nT_dev)) return;
    const int b = w.ti[k * 4];
    float* tf = w.tf + (long)k * 8;
    if (b < 0) {
        for (int i = 0; i < 8; ++i) tf[i] = 0.f;
        return;
    }
    const int a = w.ti[k * 4 + 1], gj = w.ti[k * 4 + 2], gi = w.ti[k * 4 + 3];
    const int NCH = 7 + C;
    const float* lg = logits + ((long)(b * G + gj) * G + gi) * (A * NCH) + a * NCH;
    float pb[6], tb[6];
    decode_box(lg, gi, gj, an.w[a], an.h[a], pb);
    const float* t = targets + (long)k * 8;
    const float gf = (float)G;
    tb[0] = t[2] * gf; tb[1] = t[3] * gf; tb[2] = t[4] * gf; tb[3] = t[5] * gf; tb[4] = t[6]; tb[5] = t[7];
    geom::InterPart ip;
    geom::HullPart hp;
    const float* pi = w.part + (long)k * 12;
    const float* ph = w.part + ((long)ncap + k) * 12;
    ip.inter = pi[0];
    hp.carea = GIOU ? ph[0] : 0.f; hp.sg = GIOU ? ph[1] : 0.f; hp.on = GIOU ? __float_as_int(ph[10]) : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ip.dIx[i] = pi[1 + i]; ip.dIy[i] = pi[5 + i];
        hp.dy[i] = GIOU ? ph[2 + i] : 0.f; hp.dx[i] = GIOU ? ph[6 + i] : 0.f;
    }
    const geom::PairOut o = geom::pair_finish<GIOU>(pb, tb, atan2f(pb[4], pb[5]), ip, hp);
    tf[0] = o.iou;
    tf[1] = o.term;
    for (int i = 0; i < 6; ++i) tf[2 + i] = o.g[i];
    atomicAdd(&w.acc[A_GIOU], (double)o.term);
}

template <bool GIOU>
__global__ void __launch_bounds__(64) pairs_finish_kernel(const float* __restrict__ logits, const float* __restrict__ targets, int nT,
                                                          int G, int A, int C, Anchors an, Work w) {
    pairs_finish_body<GIOU>(logits, targets, nT, nullptr, G, A, C, an, w);
}
template <bool GIOU>
__global__ void __launch_bounds__(64) pairs_finish3_kernel(HeadP h0, HeadP h1, HeadP h2, const float* __restrict__ targets, int ncap,
                                                           const int* __restrict__ nT_dev, int A, int C) {
    CY_HEAD_SWITCH(pairs_finish_body<GIOU>(h.logits, targets, ncap, nT_dev, h.G, A, C, h.an, h.w))
}

__device__ __forceinline__ float bce_grad(float p, float t) {
    // d BCE / d p as torch computes it: (p - t) / max((1-p)*p, 1e-12)
    return (p - t) / fmaxf((1.f - p) * p, 1e-12f);
}

struct Scales {
    float gx, gy, gw, gh, geul, gobj, gnoobj, gcls;  // d total / d (mean terms)
};

__device__ __forceinline__ void dense_body(const float* __restrict__ logits, const float* __restrict__ targets,
                                           int B, int G, int A, int C, const Anchors& an, const Scales& sc, const Work& w,
                                           float* __restrict__ dlogits) {
    const int NCH = 7 + C;
    const long cells = (long)B * A * G * G;
    const float nObj = (float)w.cnt[C_NOBJ];
    const float nNoobj = (float)(cells - w.cnt[C_NCLEARED]);
    double acc[A_COUNT];
#pragma unroll
    for (int i = 0; i < A_COUNT; ++i) acc[i] = 0.0;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < cells; idx += (long)gridDim.x * blockDim.x) {
        // idx enumerates (b, gj, gi, a) so that consecutive lanes read consecutive logits
        const int a = (int)(idx % A);
        const long pos = idx / A;
        const int gi = (int)(pos % G), gj = (int)((pos / G) % G), b = (int)(pos / ((long)G * G));
        const int cell = ((b * A + a) * G + gj) * G + gi;
        const float* t = logits + pos * (A * NCH) + a * NCH;
        float* d = dlogits + pos * (A * NCH) + a * NCH;
        const int own = w.owner[cell], fl = w.flags[cell];
        const float pc = sigmoidf_(t[6]);
        const float conf50 = pc > 0.5f ? 1.f : 0.f;
        acc[A_CONF50] += conf50;
        // (g[] is only ever indexed by unrolled loop counters: a run-time index into a register array would bring the VGPR-index
        // mode back, see geometry.hpp)
        float g[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) g[c] = 0.f;
        if (!(fl & 1)) {
            acc[A_BCE_NOOBJ] += -fmaxf(log1pf(-pc), -100.f);
            acc[A_CONF_NOOBJ] += pc;
            g[6] = sc.gnoobj / nNoobj * bce_grad(pc, 0.f) * pc * (1.f - pc);
        }
        if (own > 0) {
            const int k = own - 1;
            const float* tg = targets + (long)k * 8;
            const float gf = (float)G;
            const float x = tg[2] * gf, y = tg[3] * gf, tw_ = tg[4] * gf, tl_ = tg[5] * gf;
            const float tx = x - floorf(x), ty = y - floorf(y);
            const float twl = logf(tw_ / an.w[a] + 1e-16f), thl = logf(tl_ / an.h[a] + 1e-16f);
            const float sx = sigmoidf_(t[0]), sy = sigmoidf_(t[1]);
            const float ex = sx - tx, ey = sy - ty, ew = t[2] - twl, eh = t[3] - thl;
            const float eim = t[4] - tg[6], ere = t[5] - tg[7];
            const float r = sqrtf(t[4] * t[4] + t[5] * t[5]);
            acc[A_SX] += ex * ex; acc[A_SY] += ey * ey; acc[A_SW] += ew * ew; acc[A_SH] += eh * eh;
            acc[A_SIM] += eim * eim; acc[A_SRE] += ere * ere;
            acc[A_UNIT] += (1.f - r) * (1.f - r);
            acc[A_BCE_OBJ] += -fmaxf(logf(pc), -100.f);
            acc[A_CONF_OBJ] += pc;
            g[0] = sc.gx / nObj * 2.f * ex * sx * (1.f - sx);
            g[1] = sc.gy / nObj * 2.f * ey * sy * (1.f - sy);
            g[2] = sc.gw / nObj * 2.f * ew;
            g[3] = sc.gh / nObj * 2.f * eh;
            const float du = -2.f * (1.f - r) / r;
            g[4] = sc.geul / nObj * (2.f * eim + du * t[4]);
            g[5] = sc.geul / nObj * (2.f * ere + du * t[5]);
            g[6] += sc.gobj / nObj * bce_grad(pc, 1.f) * pc * (1.f - pc);
            int arg = 0;
            float bestc = -1.f;
#pragma unroll
            for (int c = 0; c < 25; ++c) {
                if (c >= C) continue;
                const float p = sigmoidf_(t[7 + c]);
                const float tc = (fl >> (8 + c)) & 1 ? 1.f : 0.f;
                acc[A_CLS] += -(tc * fmaxf(logf(p), -100.f) + (1.f - tc) * fmaxf(log1pf(-p), -100.f));
                g[7 + c] = sc.gcls / (nObj * (float)C) * bce_grad(p, tc) * p * (1.f - p);
                if (p > bestc) { bestc = p; arg = c; }
            }
            const float cmask = (arg == (int)tg[1]) ? 1.f : 0.f;
            const float iou = w.tf[(long)k * 8];
            const float det = conf50 * cmask;
            acc[A_IOU] += iou;
            acc[A_CLSACC] += cmask;
            acc[A_DET50] += (iou > 0.5f ? 1.f : 0.f) * det;
            acc[A_DET75] += (iou > 0.75f ? 1.f : 0.f) * det;
        }
#pragma unroll
        for (int c = 0; c < 32; ++c)
            if (c < NCH) d[c] = g[c];
    }
    __shared__ double red[4][A_COUNT];
#pragma unroll
    for (int i = 0; i < A_COUNT; ++i) {
        if (i == A_GIOU) continue;
        const double v = wave_sum_d(acc[i]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < A_COUNT && threadIdx.x != A_GIOU) {
        const double v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (v != 0.0) atomicAdd(&w.acc[threadIdx.x], v);
    }
}
__global__ void __launch_bounds__(256) dense_kernel(const float* __restrict__ logits, const float* __restrict__ targets,
                                                    int B, int G, int A, int C, Anchors an, Scales sc, Work w,
                                                    float* __restrict__ dlogits) {
    dense_body(logits, targets, B, G, A, C, an, sc, w, dlogits);
}
__global__ void __launch_bounds__(256) dense3_kernel(HeadP h0, HeadP h1, HeadP h2, const float* __restrict__ targets, int B, int A, int C,
                                                     Scales sc) {
    CY_HEAD_SWITCH(dense_body(h.logits, targets, B, h.G, A, C, h.an, sc, h.w, h.dlogits))
}

// d(GIoU term)/d(logits).  Targets that share a (cell, anchor) add into the same six logits.  Instead of fp32 atomics (whose
// order, and with it the last bits of d(logits), depended on what else ran on the GPU) the FIRST target of a cell sums the
// contributions of all its targets in index order and is the only writer.  The search for a cell's targets is wave-wide: 64
// candidates per __ballot instead of one dependent global load per candidate (28 us of latency per head with nT = 96).
__device__ __forceinline__ void giou_grad_body(const float* __restrict__ logits, int ncap, const int* __restrict__ nT_dev, int G, int A, int C,
                                               const Anchors& an, float lgiou, const Work& w, float* dlogits) {
    const int nT = live_rows(ncap, nT_dev);
    if (nT <= 0) return;
    const float coef = lgiou / (float)nT;      // d(mean GIoU term x its loss weight) / d term: the float32 division the host used to do
    const int lane = threadIdx.x;
    const int k = blockIdx.x * 64 + lane;
    const bool live = k < nT;
    // key of a target's (sample, anchor, row, column); rejected targets and idle lanes get keys that match nothing
    auto key_of = [&](int j) -> long {
        if (j >= nT) return -1L - j;
        const int b = w.ti[j * 4];
        if (b < 0) return -1L - j;
        return (((long)b * MAXA + w.ti[j * 4 + 1]) * 4096 + w.ti[j * 4 + 2]) * 4096 + w.ti[j * 4 + 3];
    };
    const long mine = key_of(k);
    const bool valid = live && mine >= 0;
    bool first = valid;
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float sx = 0.f, sy = 0.f, e2 = 0.f, e3 = 0.f;
    int a = 0;
    long base = 0;
    if (valid) {
        const int b = w.ti[k * 4], gj = w.ti[k * 4 + 2], gi = w.ti[k * 4 + 3];
        a = w.ti[k * 4 + 1];
        const int NCH = 7 + C;
        base = ((long)(b * G + gj) * G + gi) * (A * NCH) + a * NCH;
        const float* t = logits + base;
        sx = sigmoidf_(t[0]); sy = sigmoidf_(t[1]);
        e2 = expf(t[2]); e3 = expf(t[3]);
    }
    for (int j0 = 0; j0 < nT; j0 += 64) {
        const long other = key_of(j0 + lane);          // lane l holds candidate j0 + l
        for (int l = 0; l < 64 && j0 + l < nT; ++l) {
            const long ko = __shfl(other, l);
            const int j = j0 + l;
            if (!valid || ko != mine) continue;
            if (j < k) { first = false; continue; }     // an earlier target owns the cell's sum
            if (!first) continue;
            const float* g = w.tf + (long)j * 8 + 2;
            s[0] += coef * g[0] * sx * (1.f - sx);
            s[1] += coef * g[1] * sy * (1.f - sy);
            s[2] += e2 <= 1e3f ? coef * g[2] * e2 * an.w[a] : 0.f;
            s[3] += e3 <= 1e3f ? coef * g[3] * e3 * an.h[a] : 0.f;
            s[4] += coef * g[4];
            s[5] += coef * g[5];
        }
    }
    if (valid && first) {
#pragma unroll
        for (int i = 0; i < 6; ++i) dlogits[base + i] += s[i];
    }
}
__global__ void __launch_bounds__(64) giou_grad_kernel(const float* __restrict__ logits, int nT, int G, int A, int C, Anchors an, float lgiou,
                                                       Work w, float* dlogits) {
    giou_grad_body(logits, nT, nullptr, G, A, C, an, lgiou, w, dlogits);
}
__global__ void __launch_bounds__(64) giou_grad3_kernel(HeadP h0, HeadP h1, HeadP h2, int ncap, const int* __restrict__ nT_dev, int A, int C,
                                                        float lgiou) {
    CY_HEAD_SWITCH(giou_grad_body(h.logits, ncap, nT_dev, h.G, A, C, h.an, lgiou, h.w, h.dlogits))
}

struct LossScales {
    float noobj, obj, lgiou, leular, lobj, lcls;
};

__device__ __forceinline__ void finalize_body(const Work& w, long cells, int ncap, const int* __restrict__ nT_dev, int C, int use_giou,
                                              const LossScales& ls, float* metrics) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int nT = live_rows(ncap, nT_dev);
    const double nObj = (double)w.cnt[C_NOBJ];
    const double nNo = (double)(cells - w.cnt[C_NCLEARED]);
    const double* a = w.acc;
    const float lx = (float)(a[A_SX] / nObj), ly = (float)(a[A_SY] / nObj);
    const float lw = (float)(a[A_SW] / nObj), lh = (float)(a[A_SH] / nObj);
    const float lim = (float)(a[A_SIM] / nObj), lre = (float)(a[A_SRE] / nObj);
    const float leul = lim + lre + (float)(a[A_UNIT] / nObj);
    const float co = (float)(a[A_BCE_OBJ] / nObj), cn = (float)(a[A_BCE_NOOBJ] / nNo);
    const float lcls = (float)(a[A_CLS] / (nObj * (double)C));
    const float giou = nT > 0 ? (float)(a[A_GIOU] / (double)nT) : 0.f;
    float lobj, total;
    if (use_giou) {
        lobj = co + cn;
        total = giou * ls.lgiou + leul * ls.leular + lobj * ls.lobj + lcls * ls.lcls;
    } else {
        lobj = ls.obj * co + ls.noobj * cn;
        total = lx + ly + lw + lh + leul + lobj + lcls;
    }
    metrics[0] = total;
    metrics[1] = (float)(a[A_IOU] / nObj);
    metrics[2] = giou;
    metrics[3] = lx; metrics[4] = ly; metrics[5] = lw; metrics[6] = lh;
    metrics[7] = leul; metrics[8] = lim; metrics[9] = lre;
    metrics[10] = lobj; metrics[11] = lcls;
    metrics[12] = (float)(100.0 * a[A_CLSACC] / nObj);
    metrics[13] = (float)(a[A_DET50] / (nObj + 1e-16));
    metrics[14] = (float)(a[A_DET75] / (nObj + 1e-16));
    metrics[15] = (float)(a[A_DET50] / (a[A_CONF50] + 1e-16));
    metrics[16] = (float)(a[A_CONF_OBJ] / nObj);
    metrics[17] = (float)(a[A_CONF_NOOBJ] / nNo);
    metrics[18] = (float)nObj;
    metrics[19] = (float)w.cnt[C_ERR];
}
__global__ void finalize_kernel(Work w, long cells, int nT, int C, int use_giou, LossScales ls, float* metrics) {
    finalize_body(w, cells, nT, nullptr, C, use_giou, ls, metrics);
}
__global__ void finalize3_kernel(HeadP h0, HeadP h1, HeadP h2, int ncap, const int* __restrict__ nT_dev, int C, int use_giou, LossScales ls) {
    CY_HEAD_SWITCH(finalize_body(h.w, h.cells, ncap, nT_dev, C, use_giou, ls, h.metrics))
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

inline Work carve(void* ws, long cells, int nT, size_t* zero_bytes, size_t* total_bytes = nullptr) {
    unsigned char* p = (unsigned char*)ws;
    Work w;
    size_t off = 0;
    w.acc = (double*)(p + off); off += align_up(sizeof(double) * A_COUNT, 256);
    w.cnt = (int*)(p + off); off += 256;
    w.owner = (int*)(p + off); off += align_up(sizeof(int) * cells, 256);
    w.flags = (int*)(p + off); off += align_up(sizeof(int) * cells, 256);
    if (zero_bytes) *zero_bytes = off;
    w.ti = (int*)(p + off); off += align_up(sizeof(int) * 4 * (size_t)(nT > 0 ? nT : 1), 256);
    w.tf = (float*)(p + off); off += align_up(sizeof(float) * 8 * (size_t)(nT > 0 ? nT : 1), 256);
    w.part = (float*)(p + off); off += align_up(sizeof(float) * 24 * (size_t)(nT > 0 ? nT : 1), 256);
    if (total_bytes) *total_bytes = off;
    return w;
}

inline int fill_anchors(Anchors& an, const float* host, int A, int fields, double stride) {
    if (A < 1 || A > MAXA) return CY_ERR_ARG;
    for (int a = 0; a < MAXA; ++a) { an.w[a] = an.h[a] = 1.f; an.im[a] = 0.f; an.re[a] = 1.f; }
    for (int a = 0; a < A; ++a) {
        an.w[a] = (float)((double)host[a * fields + 0] / stride);
        an.h[a] = (float)((double)host[a * fields + 1] / stride);
        if (fields == 4) { an.im[a] = host[a * 4 + 2]; an.re[a] = host[a * 4 + 3]; }
    }
    return 0;
}

}  // namespace

extern "C" int cy_yolo_decode(const float* logits, int B, int G, int A, int C, const float* anchors_host,
                              float img_size, float* out, int rows_total, int row_offset, cy_stream_t s) {
    CY_ENTER();
    if (!logits || !out || !anchors_host || C < 1 || C > 23 || 7 + C > 32) return CY_ERR_ARG;
    Anchors an;
    const double stride = (double)img_size / (double)G;
    if (fill_anchors(an, anchors_host, A, 2, stride)) return CY_ERR_ARG;
    const long total = (long)B * G * G * A;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(decode_kernel, dim3(grid), dim3(256), 0, cy_s(s), logits, B, G, A, C, an, (float)stride, out,
                       rows_total, row_offset);
    CY_LAUNCH_CHECK();
    return 0;
}

// Scratch (private-segment) bytes per lane of the per-target kernels, as the loaded code object reports them: 0 by
// construction (geometry.hpp keeps every dynamically indexed array in LDS) and by build.py's check; the engine still asks
// before it lets the heads run beside other kernels (CY_HEADS_SIDE=1).
extern "C" int cy_head_scratch_bytes(void) {
    CY_ENTER();
    const void* fns[] = {(const void*)assign_kernel, (const void*)pairs_kernel<true>, (const void*)pairs_kernel<false>,
                         (const void*)pairs_finish_kernel<true>, (const void*)pairs_finish_kernel<false>, (const void*)giou_grad_kernel};
    int worst = 0;
    for (const void* f : fns) {
        hipFuncAttributes a;
        if (hipFuncGetAttributes(&a, f) != hipSuccess) { (void)hipGetLastError(); return -2; }
        if ((int)a.localSizeBytes > worst) worst = (int)a.localSizeBytes;
    }
    return worst;
}

extern "C" int64_t cy_yolo_loss_workspace(int B, int G, int A, int C, int nT) {
    (void)C;
    const long cells = (long)B * A * G * G;
    size_t z, total;
    (void)carve(nullptr, cells, nT, &z, &total);
    return (int64_t)total;
}

// One workspace for all the heads of a model: the regions cy_yolo_loss_multi zeroes (accumulators, counters, ownership maps of
// every head) come first and contiguous -- ONE memset -- then every head's target tables.
static size_t carve_multi(void* ws, int nheads, const int* Gs, int B, int A, int nT, Work* out, size_t* zero_bytes) {
    unsigned char* p = (unsigned char*)ws;
    size_t off = 0;
    for (int h = 0; h < nheads; ++h) {
        const long cells = (long)B * A * Gs[h] * Gs[h];
        Work& w = out[h];
        w.acc = (double*)(p + off); off += align_up(sizeof(double) * A_COUNT, 256);
        w.cnt = (int*)(p + off); off += 256;
        w.owner = (int*)(p + off); off += align_up(sizeof(int) * cells, 256);
        w.flags = (int*)(p + off); off += align_up(sizeof(int) * cells, 256);
    }
    if (zero_bytes) *zero_bytes = off;
    const size_t n1 = (size_t)(nT > 0 ? nT : 1);
    for (int h = 0; h < nheads; ++h) {
        Work& w = out[h];
        w.ti = (int*)(p + off); off += align_up(sizeof(int) * 4 * n1, 256);
        w.tf = (float*)(p + off); off += align_up(sizeof(float) * 8 * n1, 256);
        w.part = (float*)(p + off); off += align_up(sizeof(float) * 24 * n1, 256);
    }
    return off;
}

extern "C" int64_t cy_yolo_loss_multi_workspace(int nheads, const int* Gs_host, int B, int A, int C, int nT) {
    (void)C;
    if (nheads < 1 || nheads > 3 || !Gs_host) return -1;
    Work w[3];
    return (int64_t)carve_multi(nullptr, nheads, Gs_host, B, A, nT, w, nullptr);
}

// Decode + loss of up to three heads in ONE sequence of launches (head = blockIdx.y): 9 launches per step instead of 8 per head.
// The heads are a chain of one-wave kernels that is pure latency (~0.1 ms per head); since round 3 they run on the trunk's stream
// (profiles/r03_head_race.txt), where that latency is on the critical path.  Same kernels bodies, same arithmetic and the same
// results as cy_yolo_decode + cy_yolo_loss per head.
static int yolo_loss_multi_impl(int nheads, const cy_head_in* heads_host, int B, int A, int C, const float* targets, int nT,
                                const int32_t* nT_dev, float img_size, float ignore_thresh, int use_giou, void* workspace, float* out,
                                int rows_total, cy_stream_t s) {
    // nT: the row count the launches are sized for; nT_dev (optional): the live count on the device, <= nT
    if (nheads < 1 || nheads > 3 || !heads_host || !workspace || nT < 0 || (nT > 0 && !targets)) return CY_ERR_ARG;
    if (C < 1 || C > 23 || 7 + C > 32 || A < 1 || A > MAXA) return CY_ERR_ARG;
    HeadP hp[3];
    Work w[3];
    int Gs[3];
    for (int h = 0; h < nheads; ++h) {
        if (!heads_host[h].logits || !heads_host[h].dlogits || !heads_host[h].metrics || !heads_host[h].anchors_host || heads_host[h].G < 1)
            return CY_ERR_ARG;
        Gs[h] = heads_host[h].G;
    }
    size_t zero_bytes;
    (void)carve_multi(workspace, nheads, Gs, B, A, nT, w, &zero_bytes);
    long max_cells = 0;
    for (int h = 0; h < 3; ++h) {
        const int q = h < nheads ? h : 0;        // unused slots repeat head 0 (grid.y = nheads: never selected)
        const cy_head_in& in = heads_host[q];
        const double stride = (double)img_size / (double)in.G;
        HeadP& d = hp[h];
        d.logits = in.logits; d.dlogits = in.dlogits; d.metrics = in.metrics;
        d.G = in.G; d.row_offset = in.row_offset; d.stride = (float)stride;
        d.cells = (long)B * A * in.G * in.G;
        if (fill_anchors(d.an_dec, in.anchors_host, A, 4, stride) || fill_anchors(d.an, in.anchors_host, A, 4, stride)) return CY_ERR_ARG;
        d.w = w[q];
        if (d.cells > max_cells) max_cells = d.cells;
    }
    const dim3 gy(1, nheads);
    if (out) {
        const long total = max_cells;      // B * G * G * A of the largest head
        const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
        hipLaunchKernelGGL(decode3_kernel, dim3(grid, nheads), dim3(256), 0, cy_s(s), hp[0], hp[1], hp[2], B, A, C, out, rows_total);
    }
    if (hipMemsetAsync(workspace, 0, zero_bytes, cy_s(s)) != hipSuccess) return -(1000 + 1);
    const LossScales ls = {100.f, 1.f, 3.54f, 3.54f, 64.3f, 37.4f};  // reference yolo_layer.py:40-45
    Scales sc;
    if (use_giou) {
        sc.gx = sc.gy = sc.gw = sc.gh = 0.f;
        sc.geul = ls.leular; sc.gobj = ls.lobj; sc.gnoobj = ls.lobj; sc.gcls = ls.lcls;
    } else {
        sc.gx = sc.gy = sc.gw = sc.gh = 1.f;
        sc.geul = 1.f; sc.gobj = ls.obj; sc.gnoobj = ls.noobj; sc.gcls = 1.f;
    }
    const int tb = (nT + 63) / 64;
    if (nT > 0) {
        const int ta = (nT + 64 / LPT - 1) / (64 / LPT);
        hipLaunchKernelGGL(assign3_kernel, dim3(ta, nheads), dim3(64), 0, cy_s(s), hp[0], hp[1], hp[2], targets, nT, nT_dev, B, A, ignore_thresh);
        if (use_giou) {
            hipLaunchKernelGGL(pairs3_kernel<true>, dim3(2 * tb, nheads), dim3(64), 0, cy_s(s), hp[0], hp[1], hp[2], targets, nT, nT_dev, A, C, tb);
            hipLaunchKernelGGL(pairs_finish3_kernel<true>, dim3(tb, nheads), dim3(64), 0, cy_s(s), hp[0], hp[1], hp[2], targets, nT, nT_dev, A, C);
        } else {
            hipLaunchKernelGGL(pairs3_kernel<false>, dim3(tb, nheads), dim3(64), 0, cy_s(s), hp[0], hp[1], hp[2], targets, nT, nT_dev, A, C, tb);
            hipLaunchKernelGGL(pairs_finish3_kernel<false>, dim3(tb, nheads), dim3(64), 0, cy_s(s), hp[0], hp[1], hp[2], targets, nT, nT_dev, A, C);
        }
    }
    const int grid = (int)((max_cells + 255) / 256 > 2048 ? 2048 : (max_cells + 255) / 256);
    hipLaunchKernelGGL(dense3_kernel, dim3(grid, nheads), dim3(256), 0, cy_s(s), hp[0], hp[1], hp[2], targets, B, A, C, sc);
    if (nT > 0 && use_giou)
        hipLaunchKernelGGL(giou_grad3_kernel, dim3(tb, nheads), dim3(64), 0, cy_s(s), hp[0], hp[1], hp[2], nT, nT_dev, A, C, ls.lgiou);
    hipLaunchKernelGGL(finalize3_kernel, gy, dim3(64), 0, cy_s(s), hp[0], hp[1], hp[2], nT, nT_dev, C, use_giou, ls);
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_yolo_loss_multi(int nheads, const cy_head_in* heads_host, int B, int A, int C, const float* targets, int nT,
                                  float img_size, float ignore_thresh, int use_giou, void* workspace, float* out, int rows_total,
                                  cy_stream_t s) {
    CY_ENTER();
    return yolo_loss_multi_impl(nheads, heads_host, B, A, C, targets, nT, nullptr, img_size, ignore_thresh, use_giou, workspace, out,
                                rows_total, s);
}

extern "C" int cy_yolo_loss_multi_n(int nheads, const cy_head_in* heads_host, int B, int A, int C, const float* targets, int nT_cap,
                                    const int32_t* nT_dev, float img_size, float ignore_thresh, int use_giou, void* workspace,
                                    float* out, int rows_total, cy_stream_t s) {
    CY_ENTER();
    if (!nT_dev || nT_cap < 1 || !targets) return CY_ERR_ARG;
    return yolo_loss_multi_impl(nheads, heads_host, B, A, C, targets, nT_cap, nT_dev, img_size, ignore_thresh, use_giou, workspace, out,
                                rows_total, s);
}

extern "C" int cy_yolo_loss(const float* logits, int B, int G, int A, int C, const float* targets, int nT,
                            const float* anchors_host, float img_size, float ignore_thresh, int use_giou,
                            void* workspace, float* metrics, float* dlogits, cy_stream_t s) {
    CY_ENTER();
    if (!logits || !workspace || !metrics || !dlogits || !anchors_host || nT < 0 || (nT > 0 && !targets))
        return CY_ERR_ARG;
    if (C < 1 || C > 23 || 7 + C > 32) return CY_ERR_ARG;
    Anchors an;
    const double stride = (double)img_size / (double)G;
    if (fill_anchors(an, anchors_host, A, 4, stride)) return CY_ERR_ARG;
    const long cells = (long)B * A * G * G;
    size_t zero_bytes;
    Work w = carve(workspace, cells, nT, &zero_bytes);
    if (hipMemsetAsync(workspace, 0, zero_bytes, cy_s(s)) != hipSuccess) return -(1000 + 1);
    const LossScales ls = {100.f, 1.f, 3.54f, 3.54f, 64.3f, 37.4f};  // reference yolo_layer.py:40-45
    Scales sc;
    if (use_giou) {
        sc.gx = sc.gy = sc.gw = sc.gh = 0.f;
        sc.geul = ls.leular; sc.gobj = ls.lobj; sc.gnoobj = ls.lobj; sc.gcls = ls.lcls;
    } else {
        sc.gx = sc.gy = sc.gw = sc.gh = 1.f;
        sc.geul = 1.f; sc.gobj = ls.obj; sc.gnoobj = ls.noobj; sc.gcls = 1.f;
    }
    const int tb = (nT + 63) / 64;
    if (nT > 0) {
        const int ta = (nT + 64 / LPT - 1) / (64 / LPT);      // 4 lanes per target
        hipLaunchKernelGGL(assign_kernel, dim3(ta), dim3(64), 0, cy_s(s), targets, nT, B, G, A, an, ignore_thresh, w);
        if (use_giou) {      // clip blocks + hull blocks in one grid, then the join
            hipLaunchKernelGGL(pairs_kernel<true>, dim3(2 * tb), dim3(64), 0, cy_s(s), logits, targets, nT, G, A, C, an, w, tb);
            hipLaunchKernelGGL(pairs_finish_kernel<true>, dim3(tb), dim3(64), 0, cy_s(s), logits, targets, nT, G, A, C, an, w);
        } else {
            hipLaunchKernelGGL(pairs_kernel<false>, dim3(tb), dim3(64), 0, cy_s(s), logits, targets, nT, G, A, C, an, w, tb);
            hipLaunchKernelGGL(pairs_finish_kernel<false>, dim3(tb), dim3(64), 0, cy_s(s), logits, targets, nT, G, A, C, an, w);
        }
    }
    const int grid = (int)((cells + 255) / 256 > 2048 ? 2048 : (cells + 255) / 256);
    hipLaunchKernelGGL(dense_kernel, dim3(grid), dim3(256), 0, cy_s(s), logits, targets, B, G, A, C, an, sc, w, dlogits);
    if (nT > 0 && use_giou)
        hipLaunchKernelGGL(giou_grad_kernel, dim3(tb), dim3(64), 0, cy_s(s), logits, nT, G, A, C, an, ls.lgiou, w, dlogits);
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(64), 0, cy_s(s), w, cells, nT, C, use_giou, ls, metrics);
    CY_LAUNCH_CHECK();
    return 0;
}
