// LiDAR point cloud -> bird's-eye-view maps (intensity, height, density) on the device.
//
// Replaces the numpy pipeline removePoints + makeBVFeature (reference src/data_process/kitti_bev_utils.py:18-76; SURVEY.md
// section 8f row 1): range filter, z -= minZ, discretise to (xi, yi), keep per pixel the HIGHEST point (first in file
// order among equal heights -- the reference's stable lexsort by (xi, yi, -z) followed by unique(return_index)), its
// height / |maxZ - minZ|, its intensity, and min(1, log(count + 1) / log 64).
//
// HBM-bound integer/atomic work: one pass over the points (16 B each) with a 64-bit atomicMax per point on
// key = z bits << 32 | ~index (z >= 0 after the shift, so its IEEE bits order like the value) and a 32-bit count;
// one pass over the (H+1) x (W+1) pixels that decodes the key.  No sort.
#include "common.hpp"

namespace {

struct BevParams {
    float minX, maxX, minY, maxY, minZ, maxZ;
    float zshift;      // subtracted from z before rasterising (minZ for raw points, 0 for points removePoints already shifted)
    float disc;        // metres per pixel, float32 like the reference's float32-array / python-float division
    float max_height;  // the height map is z / max_height (|maxZ - minZ| of the box)
    int H, W;          // output map size (608 x 608); bins 0..H and 0..W exist, row/column H / W are dropped
};

__global__ void __launch_bounds__(256) bev_scatter_kernel(const float4* __restrict__ pts, int n, BevParams p,
                                                         unsigned long long* __restrict__ keys, unsigned* __restrict__ counts) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 q = pts[i];
    if (!(q.x >= p.minX && q.x <= p.maxX && q.y >= p.minY && q.y <= p.maxY && q.z >= p.minZ && q.z <= p.maxZ)) return;
    const float z = fmaxf(q.z - p.zshift, 0.f);   // >= 0: the key orders by the IEEE bits of z
    const int xi = (int)floorf(q.x / p.disc);
    const int yi = (int)(floorf(q.y / p.disc) + (float)(p.W + 1) / 2);   // np.int_(floor(y / d) + Width / 2): truncation
    if ((unsigned)xi >= (unsigned)p.H || (unsigned)yi >= (unsigned)p.W) return;   // bins H / W are cropped away
    const int pix = xi * p.W + yi;
    const unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
    atomicMax(keys + pix, key);
    atomicAdd(counts + pix, 1u);
}

__global__ void __launch_bounds__(256) bev_resolve_kernel(const float4* __restrict__ pts, BevParams p,
                                                         unsigned long long* __restrict__ keys, unsigned* __restrict__ counts,
                                                         float* __restrict__ out) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int npix = p.H * p.W;
    if (pix >= npix) return;
    const unsigned c = counts[pix];
    float inten = 0.f, height = 0.f, dens = 0.f;
    if (c) {
        const unsigned long long key = keys[pix];
        const unsigned idx = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
        height = __uint_as_float((unsigned)(key >> 32)) / p.max_height;
        inten = pts[idx].w;
        const double d = log((double)c + 1.0) / log(64.0);
        dens = (float)(d < 1.0 ? d : 1.0);
        keys[pix] = 0ull;      // leave the workspace zeroed for the next frame
        counts[pix] = 0u;
    }
    out[pix] = inten;
    out[npix + pix] = height;
    out[2 * npix + pix] = dens;
}

// ---- augmentation on the rasterised maps (reference kitti_dataset.py:123-173 load_mosaic, transformation.py:376-437) ----
// Pure data movement plus a few float32 operations on the target rows, written with explicit round-to-nearest
// intrinsics in the reference's order of operations (no FMA contraction) so that targets come out bit-identical.
struct MosaicRects {
    int x1a[4], y1a[4], x2a[4], y2a[4], x1b[4], y1b[4];   // destination rectangle on the canvas, source origin in the tile
};

// canvas[c][y][x] (2S x 2S) = fill, or tile i's pixel where (y, x) lies in tile i's destination rectangle.  The four
// rectangles meet at the mosaic centre and do not overlap; the scan keeps the reference's "later tile wins" order.
__global__ void __launch_bounds__(256) mosaic_kernel(const float* __restrict__ t0, const float* __restrict__ t1,
                                                    const float* __restrict__ t2, const float* __restrict__ t3, int C,
                                                    int h, int w, int S2, MosaicRects r, float fill,
                                                    float* __restrict__ out) {
    const long n = (long)C * S2 * S2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int x = (int)(i % S2), y = (int)((i / S2) % S2), c = (int)(i / ((long)S2 * S2));
        float v = fill;
#pragma unroll
        for (int k = 3; k >= 0; --k) {
            if (x >= r.x1a[k] && x < r.x2a[k] && y >= r.y1a[k] && y < r.y2a[k]) {
                const float* t = k == 0 ? t0 : (k == 1 ? t1 : (k == 2 ? t2 : t3));
                v = t[((long)c * h + (y - r.y1a[k] + r.y1b[k])) * w + (x - r.x1a[k] + r.x1b[k])];
                break;
            }
        }
        out[i] = v;
    }
}

struct MosaicPads {
    float padw[4], padh[4];
};

// targets[:, 2] = (t2 * w + padw) / (2 S); [:, 3] likewise with h, padh; [:, 4] = t4 * w / (2 S); [:, 5] = t5 * h / (2 S);
// then x, y clamped to [0, 1 - 0.5 / S]  (kitti_dataset.py:160-171)
__global__ void mosaic_targets_kernel(float* t, int nT, const int* __restrict__ tile_of, float w, float h, MosaicPads p,
                                      float S2, float cmax) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nT) return;
    float* r = t + (long)i * 8;
    const int k = tile_of[i] & 3;
    // the reference rounds after every tensor op: t * w, then + pad, then / (2 S).  hipcc contracts a float multiply
    // feeding an add into one FMA (also through __fmul_rn / __fadd_rn and through a round trip via double, which it folds
    // away first), so the product is pinned in a register by an empty asm before it is used
    auto mul32 = [](float a, float b) { float m = a * b; asm volatile("" : "+v"(m)); return m; };
    float x = (mul32(r[2], w) + p.padw[k]) / S2;      // IEEE-correct f32 division (hipcc's default for '/')
    float y = (mul32(r[3], h) + p.padh[k]) / S2;
    r[4] = mul32(r[4], w) / S2;
    r[5] = mul32(r[5], h) / S2;
    x = x < 0.f ? 0.f : (x > cmax ? cmax : x);
    y = y < 0.f ? 0.f : (y > cmax ? cmax : y);
    r[2] = x;
    r[3] = y;
}

struct CutHoles {
    int y1[8], y2[8], x1[8], x2[8];
    int n;
};

// out = horizontal flip of src (or a copy) with the hole rectangles [y1, y2) x [x1, x2) set to fill
__global__ void __launch_bounds__(256) flip_cutout_kernel(const float* __restrict__ src, int C, int H, int W, int flip,
                                                         CutHoles hs, float fill, float* __restrict__ out) {
    const long n = (long)C * H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        bool cut = false;
        for (int k = 0; k < hs.n; ++k) cut |= (y >= hs.y1[k] && y < hs.y2[k] && x >= hs.x1[k] && x < hs.x2[k]);
        out[i] = cut ? fill : src[i - x + (flip ? W - 1 - x : x)];
    }
}

// flip: x -> 1 - x, im -> -im (transformation.py:380-384).  cut-out: keep[i] = 0 for a target whose centre (x * W, y * H)
// lies inside a hole, borders included (transformation.py:428-434: x1 <= x * w <= x2 and y1 <= y * h <= y2).
__global__ void flip_cutout_targets_kernel(float* t, int nT, int flip, CutHoles hs, float W, float H, unsigned char* keep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nT) return;
    float* r = t + (long)i * 8;
    if (flip) {
        r[2] = 1.f - r[2];
        r[6] = -r[6];
    }
    if (keep) {
        float px = r[2] * W, py = r[3] * H;
        asm volatile("" : "+v"(px), "+v"(py));
        bool in = false;
        for (int k = 0; k < hs.n; ++k)
            in |= ((float)hs.x1[k] <= px && px <= (float)hs.x2[k] && (float)hs.y1[k] <= py && py <= (float)hs.y2[k]);
        keep[i] = in ? 0 : 1;
    }
}

}  // namespace

extern "C" int cy_bev_mosaic(const float* tile0, const float* tile1, const float* tile2, const float* tile3, int C, int h, int w,
                             int img_size, const int* rects_host, float fill, float* out, cy_stream_t s) {
    CY_ENTER();
    if (!tile0 || !tile1 || !tile2 || !tile3 || !rects_host || !out || C < 1 || h < 1 || w < 1 || img_size < 1) return CY_ERR_ARG;
    MosaicRects r;
    const int S2 = 2 * img_size;
    for (int k = 0; k < 4; ++k) {
        const int* q = rects_host + 6 * k;
        r.x1a[k] = q[0]; r.y1a[k] = q[1]; r.x2a[k] = q[2]; r.y2a[k] = q[3]; r.x1b[k] = q[4]; r.y1b[k] = q[5];
        // the copied window must lie inside both images
        if (q[0] < 0 || q[1] < 0 || q[2] > S2 || q[3] > S2 || q[2] < q[0] || q[3] < q[1] || q[4] < 0 || q[5] < 0 ||
            q[4] + (q[2] - q[0]) > w || q[5] + (q[3] - q[1]) > h)
            return CY_ERR_ARG;
    }
    hipLaunchKernelGGL(mosaic_kernel, dim3(2048), dim3(256), 0, cy_s(s), tile0, tile1, tile2, tile3, C, h, w, S2, r, fill, out);
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_bev_mosaic_targets(float* targets, int nT, const int32_t* tile_of_target, int h, int w, const int* pads_host,
                                     int img_size, cy_stream_t s) {
    CY_ENTER();
    if (nT < 0 || (nT > 0 && (!targets || !tile_of_target)) || !pads_host || img_size < 1) return CY_ERR_ARG;
    if (nT == 0) return 0;
    MosaicPads p;
    for (int k = 0; k < 4; ++k) { p.padw[k] = (float)pads_host[2 * k]; p.padh[k] = (float)pads_host[2 * k + 1]; }
    const float cmax = (float)(1.0 - 0.5 / (double)img_size);
    hipLaunchKernelGGL(mosaic_targets_kernel, dim3((nT + 63) / 64), dim3(64), 0, cy_s(s), targets, nT, tile_of_target, (float)w,
                       (float)h, p, (float)(2 * img_size), cmax);
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_bev_flip_cutout(const float* src, int C, int H, int W, int flip, const int* holes_host, int nholes, float fill,
                                  float* out, float* targets, int nT, uint8_t* keep, cy_stream_t s) {
    CY_ENTER();
    if (!src || !out || src == out || C < 1 || H < 1 || W < 1 || nholes < 0 || nholes > 8 || (nholes > 0 && !holes_host))
        return CY_ERR_ARG;
    if (nT < 0 || (nT > 0 && !targets)) return CY_ERR_ARG;
    CutHoles hs;
    hs.n = nholes;
    for (int k = 0; k < nholes; ++k) {
        hs.y1[k] = holes_host[4 * k]; hs.y2[k] = holes_host[4 * k + 1]; hs.x1[k] = holes_host[4 * k + 2]; hs.x2[k] = holes_host[4 * k + 3];
    }
    hipLaunchKernelGGL(flip_cutout_kernel, dim3(1024), dim3(256), 0, cy_s(s), src, C, H, W, flip, hs, fill, out);
    if (nT > 0 && (flip || keep))
        hipLaunchKernelGGL(flip_cutout_targets_kernel, dim3((nT + 63) / 64), dim3(64), 0, cy_s(s), targets, nT, flip, hs, (float)W,
                           (float)H, keep);
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t cy_bev_workspace(int H, int W) { return (int64_t)H * W * 12; }

extern "C" int cy_bev_rasterize(const float* points, int n, float minX, float maxX, float minY, float maxY, float minZ,
                                float maxZ, float zshift, float max_height, float disc, int H, int W, void* workspace,
                                float* out, cy_stream_t s) {
    CY_ENTER();
    if (!workspace || !out || H < 1 || W < 1 || n < 0 || (n > 0 && !points) || !(disc > 0.f) || !(max_height > 0.f))
        return CY_ERR_ARG;
    BevParams p;
    p.minX = minX; p.maxX = maxX; p.minY = minY; p.maxY = maxY; p.minZ = minZ; p.maxZ = maxZ;
    p.zshift = zshift; p.disc = disc; p.max_height = max_height; p.H = H; p.W = W;
    auto* keys = reinterpret_cast<unsigned long long*>(workspace);
    auto* counts = reinterpret_cast<unsigned*>(keys + (size_t)H * W);
    if (n > 0)
        hipLaunchKernelGGL(bev_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, cy_s(s),
                           reinterpret_cast<const float4*>(points), n, p, keys, counts);
    hipLaunchKernelGGL(bev_resolve_kernel, dim3((H * W + 255) / 256), dim3(256), 0, cy_s(s),
                       reinterpret_cast<const float4*>(points), p, keys, counts, out);
    CY_LAUNCH_CHECK();
    return 0;
}
