// Standalone rotated-IoU entry points and rotated NMS (greedy, and the reference's merge-NMS "v2").
// Reference: utils/iou_rotated_boxes_utils.py:64-142, utils/evaluation_utils.py:193-218, :250-276, :321-357.
// IoU decisions are taken on a float64 clip of float32 corners with a float32 tail, exactly as the
// oracle restates the reference's shapely path, so keep/suppress decisions are reproducible bit for bit.
#include "common.hpp"
#include "geometry.hpp"

namespace {

// (64-thread launch bounds, the loss variant a template parameter and build.py's promote-alloca budget keep the polygon arrays
// of these per-pair kernels in registers, as for yolo_head.hip's: no kernel of the library uses scratch memory)
template <bool GIOU>
__global__ void __launch_bounds__(64) riou_pairs_kernel(const float* __restrict__ pred, const float* __restrict__ target, int n,
                                                        float* ious, float* terms, float* gpred) {
    CY_GEOM_POOL(P);
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    float p[6], t[6];
    for (int i = 0; i < 6; ++i) { p[i] = pred[(long)k * 6 + i]; t[i] = target[(long)k * 6 + i]; }
    const geom::PairOut o = geom::pair_term_t<GIOU>(P, p, t);
    ious[k] = o.iou;
    terms[k] = o.term;
    if (gpred)
        for (int i = 0; i < 6; ++i) gpred[(long)k * 6 + i] = o.g[i];
}

__global__ void __launch_bounds__(64) riou_anchors_kernel(const float* __restrict__ anc, int nA, const float* __restrict__ tg, int nT,
                                    float* ious) {
    CY_GEOM_POOL(P);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nA * nT) return;
    const int a = idx / nT, t = idx - a * nT;
    float acx[4], acy[4], tcx[4], tcy[4];
    const float* A = anc + a * 4;
    const float* T = tg + (long)t * 4;
    geom::corners(100.f, 100.f, A[0], A[1], atan2f(A[2], A[3]), acx, acy);
    geom::corners(100.f, 100.f, T[0], T[1], atan2f(T[2], T[3]), tcx, tcy);
    ious[idx] = geom::iou_from_inter(geom::quad_inter_f64(P, acx, acy, tcx, tcy), A[0] * A[1], T[0] * T[1], 1e-16f);
}

struct BoxGeo {
    float cx[4], cy[4], area, x, y, rad;
};
__device__ __forceinline__ BoxGeo box_geo(const float* b) {
    BoxGeo g;
    geom::corners(b[0], b[1], b[2], b[3], atan2f(b[4], b[5]), g.cx, g.cy);
    g.area = b[2] * b[3];
    g.x = b[0]; g.y = b[1];
    g.rad = 0.5f * sqrtf(b[2] * b[2] + b[3] * b[3]);
    return g;
}
__device__ __forceinline__ bool far_apart(const BoxGeo& a, const BoxGeo& b) {
    const float dx = a.x - b.x, dy = a.y - b.y, r = (a.rad + b.rad) * 1.01f + 1e-3f;
    return dx * dx + dy * dy > r * r;
}
__device__ __forceinline__ float box_iou(const geom::Pool& P, const BoxGeo& a, const BoxGeo& b, float eps) {
    if (far_apart(a, b)) return 0.f;
    return geom::iou_from_inter(geom::quad_inter_f64(P, a.cx, a.cy, b.cx, b.cy), a.area, b.area, eps);
}

__global__ void __launch_bounds__(64) riou_matrix_kernel(const float* __restrict__ a, int na, const float* __restrict__ b, int nb, float eps,
                                   float* iou) {
    CY_GEOM_POOL(P);
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)na * nb) return;
    const int i = (int)(idx / nb), j = (int)(idx - (long)i * nb);
    const BoxGeo A = box_geo(a + (long)i * 6), Bx = box_geo(b + (long)j * 6);
    iou[idx] = geom::iou_from_inter(geom::quad_inter_f64(P, A.cx, A.cy, Bx.cx, Bx.cy), A.area, Bx.area, eps);
}

// ---- ranking by counting: rank_i = #{j : key_j before key_i}, keys (score desc, index asc) ---------
__global__ void __launch_bounds__(256) rank_kernel(const float* __restrict__ score, const int* __restrict__ tag,
                                                   const int* __restrict__ count_ptr, int count_fixed, int stride,
                                                   int* __restrict__ sorted_tag) {
    const int img = blockIdx.y;
    const int cnt = count_ptr ? count_ptr[img] : count_fixed;
    const float* sc = score + (long)img * stride;
    const int* tg = tag ? tag + (long)img * stride : nullptr;
    int* out = sorted_tag + (long)img * stride;
    __shared__ float tile[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= cnt) return;
    const float si = i < cnt ? sc[i] : 0.f;
    int rank = 0;
    for (int base = 0; base < cnt; base += 256) {
        __syncthreads();
        tile[threadIdx.x] = base + threadIdx.x < cnt ? sc[base + threadIdx.x] : -INFINITY;
        __syncthreads();
        const int lim = min(256, cnt - base);
        for (int j = 0; j < lim; ++j) {
            const float sj = tile[j];
            rank += (sj > si || (sj == si && base + j < i)) ? 1 : 0;
        }
    }
    if (i < cnt) out[rank] = tg ? tg[i] : i;
}

// ---- greedy NMS ----------------------------------------------------------------------------------
struct NmsWork {
    int* order;                // [B][K]
    float* geo;                // [B][K][12]: corners(8), area, x, y, rad
    float* attr;               // [B][K][10]: box(6), obj, cls_conf, cls_id, unused
    unsigned long long* mask;  // [B][K][W]
    int W;
};
inline size_t al(size_t v) { return (v + 255) / 256 * 256; }
inline NmsWork carve_nms(void* ws, int B, int K, size_t* total) {
    NmsWork w;
    unsigned char* p = (unsigned char*)ws;
    size_t off = 0;
    w.W = (K + 63) / 64;
    w.order = (int*)(p + off); off += al(sizeof(int) * (size_t)B * K);
    w.geo = (float*)(p + off); off += al(sizeof(float) * 12 * (size_t)B * K);
    w.attr = (float*)(p + off); off += al(sizeof(float) * 10 * (size_t)B * K);
    w.mask = (unsigned long long*)(p + off); off += al(sizeof(unsigned long long) * (size_t)B * K * w.W);
    if (total) *total = off;
    return w;
}

__device__ __forceinline__ BoxGeo load_geo(const float* g) {
    BoxGeo r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.cx[i] = g[i]; r.cy[i] = g[4 + i]; }
    r.area = g[8]; r.x = g[9]; r.y = g[10]; r.rad = g[11];
    return r;
}
__device__ __forceinline__ void store_geo(float* g, const BoxGeo& r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { g[i] = r.cx[i]; g[4 + i] = r.cy[i]; }
    g[8] = r.area; g[9] = r.x; g[10] = r.y; g[11] = r.rad;
}

__global__ void greedy_geo_kernel(const float* __restrict__ boxes, int K, NmsWork w) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= K) return;
    store_geo(w.geo + (long)r * 12, box_geo(boxes + (long)w.order[r] * 6));
}

// mask[b][i][word] bit j: ranked box j (> i, or >= i when SELF) overlaps ranked box i beyond thresh
template <bool SAME_CLASS>
__global__ void __launch_bounds__(64) mask_kernel(NmsWork w, int K, const int* __restrict__ counts, int count_fixed,
                                                  float thresh, float eps) {
    CY_GEOM_POOL(P);
    const int b = blockIdx.z, i = blockIdx.x, word = blockIdx.y, lane = threadIdx.x;
    const int cnt = counts ? min(counts[b], K) : count_fixed;
    if (i >= cnt || word * 64 >= cnt) return;
    unsigned long long* dst = w.mask + ((long)b * K + i) * w.W + word;
    if (word * 64 + 63 < i) {
        if (lane == 0) *dst = 0ull;
        return;
    }
    const int j = word * 64 + lane;
    bool hit = false;
    if (j < cnt && j >= i) {
        if (SAME_CLASS && j == i) {
            hit = true;  // a detection always belongs to its own group
        } else if (j > i) {
            const float* gi = w.geo + ((long)b * K + i) * 12;
            const float* gj = w.geo + ((long)b * K + j) * 12;
            bool ok = true;
            if (SAME_CLASS) ok = w.attr[((long)b * K + i) * 10 + 8] == w.attr[((long)b * K + j) * 10 + 8];
            if (ok) hit = box_iou(P, load_geo(gi), load_geo(gj), eps) > thresh;
        }
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) *dst = m;
}

__global__ void __launch_bounds__(64) greedy_sweep_kernel(NmsWork w, int K, int* __restrict__ keep, int* __restrict__ count) {
    extern __shared__ unsigned long long removed[];
    const int lane = threadIdx.x;
    for (int x = lane; x < w.W; x += 64) removed[x] = 0ull;
    __syncthreads();
    int n = 0;
    for (int i = 0; i < K; ++i) {
        const bool dead = (removed[i >> 6] >> (i & 63)) & 1ull;
        if (dead) continue;
        if (lane == 0) keep[n] = w.order[i];
        ++n;
        __syncthreads();
        const unsigned long long* row = w.mask + (long)i * w.W;
        for (int x = (i >> 6) + lane; x < w.W; x += 64) removed[x] |= row[x];
        __syncthreads();
    }
    if (lane == 0) *count = n;
}

// ---- post_processing_v2 ---------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) pp2_compact_kernel(const float* __restrict__ pred, int N, int C, float conf_thresh,
                                                           int* __restrict__ rows, float* __restrict__ scores,
                                                           int* __restrict__ counts) {
    const int b = blockIdx.x, tid = threadIdx.x, NCH = 7 + C;
    const float* P = pred + (long)b * N * NCH;
    int* R = rows + (long)b * N;
    float* S = scores + (long)b * N;
    __shared__ int wsum[16];
    __shared__ int base_s;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int start = 0; start < N; start += 1024) {
        const int r = start + tid;
        bool sel = false;
        float sc = 0.f;
        if (r < N) {
            const float* q = P + (long)r * NCH;
            sel = q[6] >= conf_thresh;
            if (sel) {
                float m = q[7];
                for (int c = 1; c < C; ++c) m = fmaxf(m, q[7 + c]);
                sc = q[6] * m;
            }
        }
        const unsigned long long bal = __ballot(sel);
        const int lane = tid & 63, wv = tid >> 6;
        const int within = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wv] = __popcll(bal);
        __syncthreads();
        int pre = 0, tot = 0;
        for (int k = 0; k < 16; ++k) { if (k < wv) pre += wsum[k]; tot += wsum[k]; }
        const int base = base_s;
        if (sel) { R[base + pre + within] = r; S[base + pre + within] = sc; }
        __syncthreads();
        if (tid == 0) base_s = base + tot;
        __syncthreads();
    }
    if (tid == 0) counts[b] = base_s;
}

__global__ void pp2_gather_kernel(const float* __restrict__ pred, int N, int C, const int* __restrict__ cand,
                                  const int* __restrict__ counts, int K, NmsWork w) {
    const int b = blockIdx.y, r = blockIdx.x * blockDim.x + threadIdx.x;
    const int cnt = min(counts[b], K);
    if (r >= cnt) return;
    const int NCH = 7 + C;
    const int row = cand[(long)b * N + r];
    const float* q = pred + ((long)b * N + row) * NCH;
    float* at = w.attr + ((long)b * K + r) * 10;
    for (int i = 0; i < 7; ++i) at[i] = q[i];
    float m = q[7];
    int arg = 0;
    for (int c = 1; c < C; ++c)
        if (q[7 + c] > m) { m = q[7 + c]; arg = c; }
    at[7] = m; at[8] = (float)arg; at[9] = 0.f;
    w.order[(long)b * K + r] = row;
    store_geo(w.geo + ((long)b * K + r) * 12, box_geo(q));
}

__global__ void __launch_bounds__(64) pp2_sweep_kernel(NmsWork w, int K, const int* __restrict__ counts,
                                                       float* __restrict__ det, int* __restrict__ det_src,
                                                       int* __restrict__ det_count) {
    extern __shared__ unsigned long long alive[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int cnt = min(counts[b], K);
    const int W = (cnt + 63) / 64;
    for (int x = lane; x < W; x += 64) {
        const int rem = cnt - x * 64;
        alive[x] = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
    }
    __syncthreads();
    int n = 0;
    for (int i = 0; i < cnt; ++i) {
        const bool live = (alive[i >> 6] >> (i & 63)) & 1ull;
        if (!live) continue;
        __syncthreads();
        const unsigned long long* row = w.mask + ((long)b * K + i) * w.W;
        float s[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int x = (i >> 6) + lane; x < W; x += 64) {
            unsigned long long grp = row[x] & alive[x];
            alive[x] &= ~grp;
            while (grp) {
                const int j = x * 64 + __builtin_ctzll(grp);
                grp &= grp - 1ull;
                const float* at = w.attr + ((long)b * K + j) * 10;
                const float wt = at[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) s[c] += wt * at[c];
                s[6] += wt;
            }
        }
#pragma unroll
        for (int c = 0; c < 7; ++c) s[c] = wave_sum(s[c]);
        if (lane == 0) {
            const float* at = w.attr + ((long)b * K + i) * 10;
            float* d = det + ((long)b * K + n) * 9;
            for (int c = 0; c < 6; ++c) d[c] = s[c] / s[6];
            d[6] = at[6]; d[7] = at[7]; d[8] = at[8];
            det_src[(long)b * K + n] = w.order[(long)b * K + i];
        }
        ++n;
        __syncthreads();
    }
    if (lane == 0) det_count[b] = n;
}

}  // namespace

extern "C" int cy_riou_pairs(const float* pred, const float* target, int n, int giou, float* ious, float* terms,
                             float* gpred, cy_stream_t s) {
    CY_ENTER();
    if (n < 0 || (n > 0 && (!pred || !target || !ious || !terms))) return CY_ERR_ARG;
    if (n == 0) return 0;
    if (giou) hipLaunchKernelGGL(riou_pairs_kernel<true>, dim3((n + 63) / 64), dim3(64), 0, cy_s(s), pred, target, n, ious, terms, gpred);
    else hipLaunchKernelGGL(riou_pairs_kernel<false>, dim3((n + 63) / 64), dim3(64), 0, cy_s(s), pred, target, n, ious, terms, gpred);
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_riou_anchors(const float* anchors_wlir, int nA, const float* targets_wlir, int nT, float* ious,
                               cy_stream_t s) {
    CY_ENTER();
    if (nA < 1 || nT < 0 || !anchors_wlir || (nT > 0 && (!targets_wlir || !ious))) return CY_ERR_ARG;
    if (nT == 0) return 0;
    hipLaunchKernelGGL(riou_anchors_kernel, dim3((nA * nT + 63) / 64), dim3(64), 0, cy_s(s), anchors_wlir, nA,
                       targets_wlir, nT, ious);
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_riou_matrix(const float* a, int na, const float* b, int nb, float eps, float* iou, cy_stream_t s) {
    CY_ENTER();
    if (na < 0 || nb < 0 || (na > 0 && nb > 0 && (!a || !b || !iou))) return CY_ERR_ARG;
    if (na == 0 || nb == 0) return 0;
    const long total = (long)na * nb;
    hipLaunchKernelGGL(riou_matrix_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, cy_s(s), a, na, b, nb, eps, iou);
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t cy_rnms_workspace(int B, int K) {
    size_t total;
    (void)carve_nms(nullptr, B < 1 ? 1 : B, K < 1 ? 1 : K, &total);
    return (int64_t)total;
}

extern "C" int cy_rnms_greedy(const float* boxes, const float* confs, int K, float nms_thresh, void* workspace,
                              int32_t* keep, int32_t* count, cy_stream_t s) {
    CY_ENTER();
    if (K < 0 || !count || (K > 0 && (!boxes || !confs || !workspace || !keep))) return CY_ERR_ARG;
    if (K == 0) return hipMemsetAsync(count, 0, sizeof(int32_t), cy_s(s)) == hipSuccess ? 0 : -(1000 + 1);
    NmsWork w = carve_nms(workspace, 1, K, nullptr);
    if ((size_t)w.W * 8 > 160 * 1024) return CY_ERR_ARG;
    hipLaunchKernelGGL(rank_kernel, dim3((K + 255) / 256, 1), dim3(256), 0, cy_s(s), confs, (const int*)nullptr,
                       (const int*)nullptr, K, K, w.order);
    hipLaunchKernelGGL(greedy_geo_kernel, dim3((K + 255) / 256), dim3(256), 0, cy_s(s), boxes, K, w);
    hipLaunchKernelGGL((mask_kernel<false>), dim3(K, w.W, 1), dim3(64), 0, cy_s(s), w, K, (const int*)nullptr, K,
                       nms_thresh, 1e-12f);
    hipLaunchKernelGGL(greedy_sweep_kernel, dim3(1), dim3(64), w.W * sizeof(unsigned long long), cy_s(s), w, K, keep,
                       count);
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_pp2_select(const float* pred, int B, int N, int C, float conf_thresh, void* workspace,
                             int32_t* cand_idx, int32_t* cand_count, cy_stream_t s) {
    CY_ENTER();
    if (!pred || !workspace || !cand_idx || !cand_count || B < 1 || N < 1 || C < 1) return CY_ERR_ARG;
    int* rows = (int*)workspace;
    float* scores = (float*)((unsigned char*)workspace + sizeof(int) * (size_t)B * N);
    hipLaunchKernelGGL(pp2_compact_kernel, dim3(B), dim3(1024), 0, cy_s(s), pred, N, C, conf_thresh, rows, scores,
                       cand_count);
    hipLaunchKernelGGL(rank_kernel, dim3((N + 255) / 256, B), dim3(256), 0, cy_s(s), scores, rows, cand_count, 0, N,
                       cand_idx);
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_pp2_merge(const float* pred, int B, int N, int C, const int32_t* cand_idx,
                            const int32_t* cand_count, int Kmax, float nms_thresh, void* workspace, float* det,
                            int32_t* det_src, int32_t* det_count, cy_stream_t s) {
    CY_ENTER();
    if (!pred || !cand_idx || !cand_count || !workspace || !det || !det_src || !det_count || Kmax < 1) return CY_ERR_ARG;
    NmsWork w = carve_nms(workspace, B, Kmax, nullptr);
    if ((size_t)w.W * 8 > 160 * 1024) return CY_ERR_ARG;
    hipLaunchKernelGGL(pp2_gather_kernel, dim3((Kmax + 255) / 256, B), dim3(256), 0, cy_s(s), pred, N, C, cand_idx,
                       cand_count, Kmax, w);
    hipLaunchKernelGGL((mask_kernel<true>), dim3(Kmax, w.W, B), dim3(64), 0, cy_s(s), w, Kmax, cand_count, 0,
                       nms_thresh, 1e-16f);
    hipLaunchKernelGGL(pp2_sweep_kernel, dim3(B), dim3(64), w.W * sizeof(unsigned long long), cy_s(s), w, Kmax,
                       cand_count, det, det_src, det_count);
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_version(void) {
    CY_ENTER(); return 100; }

extern "C" int cy_device_info(int* cus, int* wave) {
    CY_ENTER();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -(1000 + 1);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -(1000 + 1);
    if (cus) *cus = prop.multiProcessorCount;
    if (wave) *wave = prop.warpSize;
    return 0;
}
