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

}  // namespace

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
