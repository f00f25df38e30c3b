// HBM-bound passes of the conv stack: BatchNorm statistics / apply / backward, activations, shortcut,
// max-pool, nearest upsample, channel-slice copies, layout packing.  All NHWC views, 16-byte chunks
// per lane, per-channel parameters held in registers (a thread always works on the same channel chunk).
#include "common.hpp"

namespace {

// A block walks pixels [blockIdx*ppb, ...) ; thread = (row = tid / cpr, chunk = tid % cpr).
// cpr = C / CH must divide 256 (true for every BN layer: C in {32..1024}, f16/f32).
struct RowMap {
    int cpr, rpp;  // chunks per row, rows per pass
};

template <typename T, int ACT, bool RES>
__global__ void __launch_bounds__(256) bn_act_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy,
                                                        const T* __restrict__ res, int ldres, long M, int C,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int ppb) {
    constexpr int CH = Elem<T>::CH;
    const int cpr = C / CH, rpp = 256 / cpr;
    const int chunk = threadIdx.x % cpr, row = threadIdx.x / cpr;
    float sc[CH], sh[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) { sc[i] = scale[chunk * CH + i]; sh[i] = shift[chunk * CH + i]; }
    const long p0 = (long)blockIdx.x * ppb;
    const long p1 = p0 + ppb < M ? p0 + ppb : M;
    for (long p = p0 + row; p < p1; p += rpp) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + p * ldx + chunk * CH);
        float f[CH];
        chunk_to_f32<T>(v, f);
        float r[CH];
        if (RES) {
            const u32x4 rv = *reinterpret_cast<const u32x4*>(res + p * ldres + chunk * CH);
            chunk_to_f32<T>(rv, r);
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            f[i] = act_f<ACT, sizeof(T) == 2>(f[i] * sc[i] + sh[i]);
            if (RES) f[i] += r[i];
        }
        *reinterpret_cast<u32x4*>(y + p * ldy + chunk * CH) = f32_to_chunk<T>(f);
    }
}

template <typename T, int ACT>
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy,
                                                           int lddy, long M, int C, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int ppb,
                                                           float* __restrict__ part, int det) {
    constexpr int CH = Elem<T>::CH;
    __shared__ float red[256 * CH * 2];
    const int cpr = C / CH, rpp = 256 / cpr;
    const int chunk = threadIdx.x % cpr, row = threadIdx.x / cpr;
    float sc[CH], sh[CH], mu[CH], is[CH], s1[CH], s2[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = chunk * CH + i;
        sc[i] = scale[c]; sh[i] = shift[c]; mu[i] = mean[c]; is[i] = invstd[c];
        s1[i] = 0.f; s2[i] = 0.f;
    }
    const long p0 = (long)blockIdx.x * ppb;
    const long p1 = p0 + ppb < M ? p0 + ppb : M;
    for (long p = p0 + row; p < p1; p += rpp) {
        float f[CH], g[CH];
        chunk_to_f32<T>(*reinterpret_cast<const u32x4*>(x + p * ldx + chunk * CH), f);
        chunk_to_f32<T>(*reinterpret_cast<const u32x4*>(dy + p * lddy + chunk * CH), g);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const float dz = g[i] * act_grad<ACT, sizeof(T) == 2>(f[i] * sc[i] + sh[i]);
            s1[i] += dz;
            s2[i] += dz * (f[i] - mu[i]) * is[i];
        }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        red[(threadIdx.x * 2 + 0) * CH + i] = s1[i];
        red[(threadIdx.x * 2 + 1) * CH + i] = s2[i];
    }
    __syncthreads();
    // thread t < cpr*CH*2 sums column (chunk, which, i) over the rpp rows
    for (int o = threadIdx.x; o < cpr * CH * 2; o += 256) {
        const int i = o % CH, which = (o / CH) & 1, ck = o / (2 * CH);
        float s = 0.f;
        for (int r = 0; r < rpp; ++r) s += red[((r * cpr + ck) * 2 + which) * CH + i];
        atomicAdd(&part[((size_t)(det ? blockIdx.x : (blockIdx.x & (CY_STAT_BINS - 1))) * 2 + which) * C + ck * CH + i], s);
    }
}

template <typename T, int ACT>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const T* __restrict__ x, int ldx, const T* dy, int lddy,
                                                          T* dx, int lddx, T* resg, int ldrg, int res_accum, long M,
                                                          int C, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift,
                                                          const float* __restrict__ dgs,
                                                          const float* __restrict__ dbs, int ppb) {
    constexpr int CH = Elem<T>::CH;
    const int cpr = C / CH, rpp = 256 / cpr;
    const int chunk = threadIdx.x % cpr, row = threadIdx.x / cpr;
    const float invM = 1.f / (float)M;
    float sc[CH], sh[CH], mu[CH], is[CH], mg[CH], mb[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = chunk * CH + i;
        sc[i] = scale[c]; sh[i] = shift[c]; mu[i] = mean[c]; is[i] = invstd[c];
        mg[i] = dgs[c] * invM; mb[i] = dbs[c] * invM;
    }
    const long p0 = (long)blockIdx.x * ppb;
    const long p1 = p0 + ppb < M ? p0 + ppb : M;
    for (long p = p0 + row; p < p1; p += rpp) {
        float f[CH], g[CH];
        chunk_to_f32<T>(*reinterpret_cast<const u32x4*>(x + p * ldx + chunk * CH), f);
        const u32x4 gv = *reinterpret_cast<const u32x4*>(dy + p * lddy + chunk * CH);
        chunk_to_f32<T>(gv, g);
        if (resg) {
            if (res_accum) {
                float o[CH];
                chunk_to_f32<T>(*reinterpret_cast<const u32x4*>(resg + p * ldrg + chunk * CH), o);
#pragma unroll
                for (int i = 0; i < CH; ++i) o[i] += g[i];
                *reinterpret_cast<u32x4*>(resg + p * ldrg + chunk * CH) = f32_to_chunk<T>(o);
            } else {
                *reinterpret_cast<u32x4*>(resg + p * ldrg + chunk * CH) = gv;
            }
        }
        float o[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const float dz = g[i] * act_grad<ACT, sizeof(T) == 2>(f[i] * sc[i] + sh[i]);
            const float xh = (f[i] - mu[i]) * is[i];
            o[i] = sc[i] * (dz - mb[i] - xh * mg[i]);
        }
        *reinterpret_cast<u32x4*>(dx + p * lddx + chunk * CH) = f32_to_chunk<T>(o);
    }
}

// ---- fused finalisers ------------------------------------------------------------------------------------------
// The statistics fold (cy_bn_finalize / cy_bn_bwd_finalize: 214 launches of ~5 us per step, each a dependent kernel
// boundary on the critical path) moved into the prologue of the kernel that consumes its result.  A block covers a group
// of CG <= 128 channels (grid.y) and a range of pixels (grid.x), so its fold is CY_STAT_BINS x 2 x CG floats (<= 16 KB from
// L2).  The first pixel block of every channel group also writes what the finaliser used to write (mean / invstd / scale /
// shift and the running statistics; the parameter gradients in the backward kernel).  The table cannot be zeroed by the
// kernel that reads it (other blocks may still be reading), so two tables alternate from one BatchNorm layer to the next
// and every launch zeroes the OTHER one (last read by the previous layer's kernel, next written by the following conv).
struct BnFoldParams {
    const float* bins;       // [CY_STAT_BINS][2][bins_ld], this layer's channels from column bins_c0 (a fused conv of two sibling
                             // layers leaves ONE table of Ca + Cb channels: cy_bn_act_fwd_fused's stats_ld / stats_c0)
    int bins_ld, bins_c0;
    float* zero_table;       // the other table of the pair
    int zero_n;
    double count;
    const float* gamma;
    const float* beta;
    float* rmean;
    float* rvar;
    long long* nbt;
    float momentum, eps;
    float* vec;              // [4][vec_ld]: mean, invstd, scale, shift (vec_ld = C, or the width of a table two layers share)
    int vec_ld;
};

__device__ __forceinline__ void zero_other_table(float* t, int n) {
    const int nthreads = gridDim.x * gridDim.y * blockDim.x;
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += nthreads) t[i] = 0.f;
}

template <typename T, int ACT, bool RES>
__global__ void __launch_bounds__(256) bn_act_fwd_fused_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy,
                                                              const T* __restrict__ res, int ldres, long M, int C, int CG,
                                                              BnFoldParams f, int ppb) {
    constexpr int CH = Elem<T>::CH;
    __shared__ float s_sc[128], s_sh[128];
    const int c0 = blockIdx.y * CG;
    // fold: thread t < CG owns channel c0 + t; bins in index order, double accumulation (what cy_bn_finalize does)
    if ((int)threadIdx.x < CG) {
        const int c = c0 + threadIdx.x;
        double s = 0.0, q = 0.0;
#pragma unroll
        for (int b = 0; b < CY_STAT_BINS; ++b) {
            s += (double)f.bins[((size_t)b * 2) * f.bins_ld + f.bins_c0 + c];
            q += (double)f.bins[((size_t)b * 2 + 1) * f.bins_ld + f.bins_c0 + c];
        }
        const double m = s / f.count;
        double var = q / f.count - m * m;
        if (var < 0.0) var = 0.0;
        const float is = (float)(1.0 / sqrt(var + (double)f.eps));
        const float sc = f.gamma[c] * is;
        const float sh = f.beta[c] - (float)m * sc;
        s_sc[threadIdx.x] = sc;
        s_sh[threadIdx.x] = sh;
        if (blockIdx.x == 0) {
            f.vec[c] = (float)m;
            f.vec[f.vec_ld + c] = is;
            f.vec[2 * f.vec_ld + c] = sc;
            f.vec[3 * f.vec_ld + c] = sh;
            if (f.rmean) {
                const double unb = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
                f.rmean[c] = (1.f - f.momentum) * f.rmean[c] + f.momentum * (float)m;
                f.rvar[c] = (1.f - f.momentum) * f.rvar[c] + f.momentum * (float)unb;
            }
            if (f.nbt && c == 0) *f.nbt += 1;
        }
    }
    zero_other_table(f.zero_table, f.zero_n);
    __syncthreads();
    const int cpr = CG / CH, rpp = 256 / cpr;
    const int chunk = threadIdx.x % cpr, row = threadIdx.x / cpr;
    float sc[CH], sh[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) { sc[i] = s_sc[chunk * CH + i]; sh[i] = s_sh[chunk * CH + i]; }
    const long p0 = (long)blockIdx.x * ppb;
    const long p1 = p0 + ppb < M ? p0 + ppb : M;
    const int cc = c0 + chunk * CH;
    for (long p = p0 + row; p < p1; p += rpp) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + p * ldx + cc);
        float fv[CH];
        chunk_to_f32<T>(v, fv);
        float r[CH];
        if (RES) {
            const u32x4 rv = *reinterpret_cast<const u32x4*>(res + p * ldres + cc);
            chunk_to_f32<T>(rv, r);
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            fv[i] = act_f<ACT, sizeof(T) == 2>(fv[i] * sc[i] + sh[i]);
            if (RES) fv[i] += r[i];
        }
        *reinterpret_cast<u32x4*>(y + p * ldy + cc) = f32_to_chunk<T>(fv);
    }
}

struct BnBwdFoldParams {
    const float* bins;       // [CY_STAT_BINS][2][bins_ld]: sum dz, sum dz * xhat; this layer's channels from column bins_c0 (one dgrad
                             // launch that last writes the output gradients of TWO layers -- a CSP concatenation -- leaves one table)
    int bins_ld, bins_c0;
    float* zero_table;
    int zero_n;
    float* ggamma;
    float* gbeta;
    float gscale;
};

template <typename T, int ACT>
__global__ void __launch_bounds__(256) bn_bwd_apply_fused_kernel(const T* __restrict__ x, int ldx, const T* dy, int lddy, T* dx,
                                                                int lddx, T* resg, int ldrg, int res_accum, long M, int C,
                                                                int CG, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift, BnBwdFoldParams f, int ppb) {
    constexpr int CH = Elem<T>::CH;
    __shared__ float s_dg[128], s_db[128];
    const int c0 = blockIdx.y * CG;
    if ((int)threadIdx.x < CG) {
        const int c = c0 + threadIdx.x;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int b = 0; b < CY_STAT_BINS; ++b) {
            s1 += (double)f.bins[((size_t)b * 2) * f.bins_ld + f.bins_c0 + c];
            s2 += (double)f.bins[((size_t)b * 2 + 1) * f.bins_ld + f.bins_c0 + c];
        }
        s_db[threadIdx.x] = (float)s1;
        s_dg[threadIdx.x] = (float)s2;
        if (blockIdx.x == 0) {
            if (f.gbeta) f.gbeta[c] += f.gscale * (float)s1;
            if (f.ggamma) f.ggamma[c] += f.gscale * (float)s2;
        }
    }
    zero_other_table(f.zero_table, f.zero_n);
    __syncthreads();
    const int cpr = CG / CH, rpp = 256 / cpr;
    const int chunk = threadIdx.x % cpr, row = threadIdx.x / cpr;
    const float invM = 1.f / (float)M;
    float sc[CH], sh[CH], mu[CH], is[CH], mg[CH], mb[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = c0 + chunk * CH + i;
        sc[i] = scale[c]; sh[i] = shift[c]; mu[i] = mean[c]; is[i] = invstd[c];
        mg[i] = s_dg[chunk * CH + i] * invM; mb[i] = s_db[chunk * CH + i] * invM;
    }
    const long p0 = (long)blockIdx.x * ppb;
    const long p1 = p0 + ppb < M ? p0 + ppb : M;
    const int cc = c0 + chunk * CH;
    for (long p = p0 + row; p < p1; p += rpp) {
        float fv[CH], g[CH];
        chunk_to_f32<T>(*reinterpret_cast<const u32x4*>(x + p * ldx + cc), fv);
        const u32x4 gv = *reinterpret_cast<const u32x4*>(dy + p * lddy + cc);
        chunk_to_f32<T>(gv, g);
        if (resg) {
            if (res_accum) {
                float o[CH];
                chunk_to_f32<T>(*reinterpret_cast<const u32x4*>(resg + p * ldrg + cc), o);
#pragma unroll
                for (int i = 0; i < CH; ++i) o[i] += g[i];
                *reinterpret_cast<u32x4*>(resg + p * ldrg + cc) = f32_to_chunk<T>(o);
            } else {
                *reinterpret_cast<u32x4*>(resg + p * ldrg + cc) = gv;
            }
        }
        float o[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const float dz = g[i] * act_grad<ACT, sizeof(T) == 2>(fv[i] * sc[i] + sh[i]);
            const float xh = (fv[i] - mu[i]) * is[i];
            o[i] = sc[i] * (dz - mb[i] - xh * mg[i]);
        }
        *reinterpret_cast<u32x4*>(dx + p * lddx + cc) = f32_to_chunk<T>(o);
    }
}

// finish: one wave per channel, lane k owns bin k of the [64][2][C] fp32 table (filled with atomics by the conv
// epilogue / the backward reduce); the lane that read a bin zeroes it, so the table is clean for its next user.
constexpr int CY_BINS = CY_STAT_BINS;
__global__ void __launch_bounds__(256) bn_finalize_kernel(float* __restrict__ bins, int rows, int C, double count,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* rmean, float* rvar, long long* nbt, float momentum,
                                                         float eps, float* mean, float* invstd, float* scale, float* shift) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    // lane k folds rows k, k + 64, ... in that order, then a fixed butterfly over the lanes: the result does not depend
    // on which block produced which row when (the deterministic mode has one row per pixel tile)
    double s = 0.0, q = 0.0;
    for (int r = lane; r < rows; r += 64) {
        float* p0 = bins + ((size_t)r * 2) * C + c;
        float* p1 = bins + ((size_t)r * 2 + 1) * C + c;
        s += (double)*p0;
        q += (double)*p1;
        *p0 = 0.f;
        *p1 = 0.f;
    }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    if (lane != 0) return;
    const double m = s / count;
    double var = q / count - m * m;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)m;
    invstd[c] = is;
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - (float)m * sc;
    if (rmean) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
    if (nbt && c == 0) *nbt += 1;
}

__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, int C,
                                      float eps, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

__global__ void __launch_bounds__(256) bn_bwd_finalize_kernel(float* __restrict__ bins, int rows, int C, float* dgs, float* dbs,
                                                             float* ggamma, float* gbeta, float gscale) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int r = lane; r < rows; r += 64) {
        float* p0 = bins + ((size_t)r * 2) * C + c;
        float* p1 = bins + ((size_t)r * 2 + 1) * C + c;
        s1 += (double)*p0;
        s2 += (double)*p1;
        *p0 = 0.f;
        *p1 = 0.f;
    }
    s1 = wave_sum_d(s1);
    s2 = wave_sum_d(s2);
    if (lane != 0) return;
    dbs[c] = (float)s1;
    dgs[c] = (float)s2;
    if (gbeta) gbeta[c] += gscale * (float)s1;
    if (ggamma) ggamma[c] += gscale * (float)s2;
}

// ---- generic chunked element kernels (any C multiple of CH) ------------------------------------
template <typename T, int MODE>  // 0: y = x, 1: y += x, 2: y = a + b (x = a, z = b)
__global__ void slice_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ z, int ldz, T* y, int ldy, long M,
                             int C) {
    constexpr int CH = Elem<T>::CH;
    const int cpr = C / CH;
    const long total = M * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / cpr;
        const int ck = (int)(i - p * cpr);
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + p * ldx + ck * CH);
        if (MODE == 0) {
            *reinterpret_cast<u32x4*>(y + p * ldy + ck * CH) = v;
        } else {
            float a[CH], b[CH];
            chunk_to_f32<T>(v, a);
            if (MODE == 1) chunk_to_f32<T>(*reinterpret_cast<const u32x4*>(y + p * ldy + ck * CH), b);
            else chunk_to_f32<T>(*reinterpret_cast<const u32x4*>(z + p * ldz + ck * CH), b);
#pragma unroll
            for (int e = 0; e < CH; ++e) a[e] += b[e];
            *reinterpret_cast<u32x4*>(y + p * ldy + ck * CH) = f32_to_chunk<T>(a);
        }
    }
}

// Max pooling as two separable passes (rows, then columns).  Exactly torch's result including its tie rule (first
// maximum in row-major window order = first row that holds the maximum, first column inside that row), with k + k loads
// per output instead of k * k (SPP: k = 5 / 9 / 13).  The per-pass tap indices are kept for the backward pass, which
// gathers through the same two stages instead of scattering with atomics (a local maximum is the arg-max of up to k*k
// windows: that many same-address atomics).
template <int CH>
__device__ __forceinline__ void store_bytes(uint8_t* dst, const int* v) {
    if constexpr (CH == 8) {
        unsigned long long w = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) w |= (unsigned long long)(v[e] & 0xFF) << (8 * e);
        *reinterpret_cast<unsigned long long*>(dst) = w;
    } else {
        unsigned w = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) w |= (unsigned)(v[e] & 0xFF) << (8 * e);
        *reinterpret_cast<unsigned*>(dst) = w;
    }
}
template <int CH>
__device__ __forceinline__ void load_bytes(const uint8_t* src, int* v) {
    if constexpr (CH == 8) {
        const unsigned long long w = *reinterpret_cast<const unsigned long long*>(src);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (int)((w >> (8 * e)) & 0xFF);
    } else {
        const unsigned w = *reinterpret_cast<const unsigned*>(src);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (int)((w >> (8 * e)) & 0xFF);
    }
}

// pass 1: m1[n][h][ow] = max_b x[n][h][ow*stride - pad + b], tap index b1
template <typename T>
__global__ void maxpool_rows_kernel(const T* __restrict__ x, int N, int H, int W, int C, int ldx, T* __restrict__ m1, int OW,
                                    int k, int stride, int pad, uint8_t* __restrict__ b1) {
    constexpr int CH = Elem<T>::CH;
    const int cpr = C / CH;
    const long total = (long)N * H * OW * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / cpr;                       // (n, h, ow)
        const int ck = (int)(i - p * cpr);
        const int ow = (int)(p % OW);
        const long nh = p / OW;                       // n * H + h
        float best[CH];
        int arg[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) { best[e] = -INFINITY; arg[e] = 0; }
        // (the tap range is clamped to the row instead of skipping taps inside the loop: a branch-free body lets the loads of
        // several taps fly together -- the k = 13 pool of the SPP block used to pay 13 memory latencies back to back)
        const int w0 = ow * stride - pad;
        const int b_lo = w0 < 0 ? -w0 : 0, b_hi = W - w0 < k ? W - w0 : k;
#pragma unroll 4
        for (int b = b_lo; b < b_hi; ++b) {
            const int w = w0 + b;
            float f[CH];
            chunk_to_f32<T>(*reinterpret_cast<const u32x4*>(x + (nh * W + w) * ldx + ck * CH), f);
#pragma unroll
            for (int e = 0; e < CH; ++e)
                if (f[e] > best[e] || f[e] != f[e]) { best[e] = f[e]; arg[e] = b; }
        }
        *reinterpret_cast<u32x4*>(m1 + p * C + ck * CH) = f32_to_chunk<T>(best);
        if (b1) store_bytes<CH>(b1 + p * C + ck * CH, arg);
    }
}

// pass 2: y[n][oh][ow] = max_a m1[n][oh*stride - pad + a][ow], tap index a1
template <typename T>
__global__ void maxpool_cols_kernel(const T* __restrict__ m1, int N, int H, int C, T* __restrict__ y, int OH, int OW, int ldy,
                                    int k, int stride, int pad, uint8_t* __restrict__ a1) {
    constexpr int CH = Elem<T>::CH;
    const int cpr = C / CH;
    const long total = (long)N * OH * OW * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / cpr;                       // (n, oh, ow)
        const int ck = (int)(i - p * cpr);
        const int ow = (int)(p % OW), oh = (int)((p / OW) % OH), n = (int)(p / ((long)OW * OH));
        float best[CH];
        int arg[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) { best[e] = -INFINITY; arg[e] = 0; }
        const int h0 = oh * stride - pad;
        const int a_lo = h0 < 0 ? -h0 : 0, a_hi = H - h0 < k ? H - h0 : k;
#pragma unroll 4
        for (int a = a_lo; a < a_hi; ++a) {
            const int h = h0 + a;
            float f[CH];
            chunk_to_f32<T>(*reinterpret_cast<const u32x4*>(m1 + (((long)n * H + h) * OW + ow) * C + ck * CH), f);
#pragma unroll
            for (int e = 0; e < CH; ++e)
                if (f[e] > best[e] || f[e] != f[e]) { best[e] = f[e]; arg[e] = a; }
        }
        *reinterpret_cast<u32x4*>(y + p * ldy + ck * CH) = f32_to_chunk<T>(best);
        if (a1) store_bytes<CH>(a1 + p * C + ck * CH, arg);
    }
}

// backward pass 1: dm1[n][r][ow] = sum over the outputs (oh, ow) whose column pass selected row r
template <typename T, bool S1>      // S1: stride 1 (the SPP pools), no divisibility test per tap
__global__ void maxpool_bwd_cols_kernel(const T* __restrict__ dy, int N, int OH, int OW, int C, int lddy,
                                        const uint8_t* __restrict__ a1, float* __restrict__ dm1, int H, int k, int stride,
                                        int pad) {
    constexpr int CH = Elem<T>::CH;
    const int cpr = C / CH;
    const long total = (long)N * H * OW * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / cpr;                       // (n, r, ow)
        const int ck = (int)(i - p * cpr);
        const int ow = (int)(p % OW), r = (int)((p / OW) % H), n = (int)(p / ((long)OW * H));
        float acc[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) acc[e] = 0.f;
        // taps whose output row exists: 0 <= t = r + pad - a < OH * stride
        const int a_lo = r + pad - (OH * stride - 1) > 0 ? r + pad - (OH * stride - 1) : 0;
        const int a_hi = r + pad < k - 1 ? r + pad : k - 1;
#pragma unroll 4
        for (int a = a_lo; a <= a_hi; ++a) {
            const int t = r + pad - a;
            if (!S1 && t % stride) continue;
            const int oh = S1 ? t : t / stride;
            const long q = ((long)n * OH + oh) * OW + ow;
            int sel[CH];
            load_bytes<CH>(a1 + q * C + ck * CH, sel);
            float f[CH];
            chunk_to_f32<T>(*reinterpret_cast<const u32x4*>(dy + q * lddy + ck * CH), f);
#pragma unroll
            for (int e = 0; e < CH; ++e)
                if (sel[e] == a) acc[e] += f[e];
        }
        float* dst = dm1 + p * C + ck * CH;
#pragma unroll
        for (int e = 0; e < CH; ++e) dst[e] = acc[e];
    }
}

// backward pass 2: dx[n][r][w] (+)= sum over the (r, ow) whose row pass selected column w
template <typename T, bool S1>
__global__ void maxpool_bwd_rows_kernel(const float* __restrict__ dm1, const uint8_t* __restrict__ b1, int N, int H, int OW,
                                        int C, T* __restrict__ dx, int W, int lddx, int k, int stride, int pad,
                                        int accumulate) {
    constexpr int CH = Elem<T>::CH;
    const int cpr = C / CH;
    const long total = (long)N * H * W * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / cpr;                       // (n, r, w)
        const int ck = (int)(i - p * cpr);
        const int w = (int)(p % W);
        const long nr = p / W;                        // n * H + r
        float acc[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) acc[e] = 0.f;
        const int b_lo = w + pad - (OW * stride - 1) > 0 ? w + pad - (OW * stride - 1) : 0;
        const int b_hi = w + pad < k - 1 ? w + pad : k - 1;
#pragma unroll 4
        for (int b = b_lo; b <= b_hi; ++b) {
            const int t = w + pad - b;
            if (!S1 && t % stride) continue;
            const int ow = S1 ? t : t / stride;
            const long q = (nr * OW + ow) * C + ck * CH;
            int sel[CH];
            load_bytes<CH>(b1 + q, sel);
            float g[CH];     // whole 16-byte groups, unconditionally: the selects below cost nothing beside a dependent load
#pragma unroll
            for (int e4 = 0; e4 < CH; e4 += 4) {
                const float4 v = *reinterpret_cast<const float4*>(dm1 + q + e4);
                g[e4] = v.x; g[e4 + 1] = v.y; g[e4 + 2] = v.z; g[e4 + 3] = v.w;
            }
#pragma unroll
            for (int e = 0; e < CH; ++e)
                if (sel[e] == b) acc[e] += g[e];
        }
        T* dst = dx + p * lddx + ck * CH;
        if (accumulate) {
            float old[CH];
            chunk_to_f32<T>(*reinterpret_cast<const u32x4*>(dst), old);
#pragma unroll
            for (int e = 0; e < CH; ++e) acc[e] += old[e];
        }
        *reinterpret_cast<u32x4*>(dst) = f32_to_chunk<T>(acc);
    }
}

template <typename T>
__global__ void f32_to_view_kernel(const float* __restrict__ x, long M, int C, float scale,
                                   const float* __restrict__ scale_dev, T* __restrict__ y, int ldy, int CPad,
                                   int accumulate) {
    const long total = M * CPad;
    if (scale_dev) scale *= *scale_dev;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / CPad;
        const int c = (int)(i - p * CPad);
        float v = c < C ? x[p * C + c] * scale : 0.f;
        if (accumulate) v += (float)y[p * ldy + c];
        y[p * ldy + c] = (T)v;
    }
}

template <typename T>
__global__ void upsample_fwd_kernel(const T* __restrict__ x, int N, int H, int W, int C, int ldx, T* __restrict__ y,
                                    int ldy, int st) {
    constexpr int CH = Elem<T>::CH;
    const int cpr = C / CH, OH = H * st, OW = W * st;
    const long total = (long)N * OH * OW * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / cpr;
        const int ck = (int)(i - p * cpr);
        const int ow = (int)(p % OW), oh = (int)((p / OW) % OH), n = (int)(p / ((long)OW * OH));
        *reinterpret_cast<u32x4*>(y + p * ldy + ck * CH) =
            *reinterpret_cast<const u32x4*>(x + ((long)(n * H + oh / st) * W + ow / st) * ldx + ck * CH);
    }
}

template <typename T>
__global__ void upsample_bwd_kernel(const T* __restrict__ dy, int N, int H, int W, int C, int lddy, T* dx, int lddx,
                                    int st, int accumulate) {
    constexpr int CH = Elem<T>::CH;
    const int cpr = C / CH, OW = W * st;
    const long total = (long)N * H * W * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / cpr;
        const int ck = (int)(i - p * cpr);
        const int w = (int)(p % W), h = (int)((p / W) % H), n = (int)(p / ((long)W * H));
        float s[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) s[e] = 0.f;
        if (accumulate) chunk_to_f32<T>(*reinterpret_cast<const u32x4*>(dx + p * lddx + ck * CH), s);
        for (int a = 0; a < st; ++a)
            for (int b = 0; b < st; ++b) {
                float f[CH];
                const long q = ((long)(n * H * st + h * st + a)) * OW + w * st + b;
                chunk_to_f32<T>(*reinterpret_cast<const u32x4*>(dy + q * lddy + ck * CH), f);
#pragma unroll
                for (int e = 0; e < CH; ++e) s[e] += f[e];
            }
        *reinterpret_cast<u32x4*>(dx + p * lddx + ck * CH) = f32_to_chunk<T>(s);
    }
}

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, int N, int C, int H, int W, int CPad,
                                    T* __restrict__ y) {
    const long total = (long)N * H * W;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        const long hw = (long)H * W;
        const long n = p / hw, r = p - n * hw;
        if (CPad * (int)sizeof(T) == 16 && C <= 8) {
            // the usual case (3 BEV channels padded to one 16-byte chunk): the planes' loads in flight together, one store
            float f[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) f[c] = c < C ? x[(n * C + c) * hw + r] : 0.f;
            T v[16 / sizeof(T)];
#pragma unroll
            for (int c = 0; c < (int)(16 / sizeof(T)); ++c) v[c] = (T)f[c];
            *reinterpret_cast<u32x4*>(y + p * CPad) = *reinterpret_cast<const u32x4*>(v);
            continue;
        }
        for (int c = 0; c < CPad; ++c) y[p * CPad + c] = c < C ? (T)x[(n * C + c) * hw + r] : (T)0.f;
    }
}

template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, int Co, int Ci, int ks, int CoPad, int CiPad,
                                    T* __restrict__ wf, T* __restrict__ wd) {
    const int kk = ks * ks;
    const long total = (long)CoPad * kk * CiPad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % CiPad);
        const int tap = (int)((i / CiPad) % kk);
        const int co = (int)(i / ((long)CiPad * kk));
        const float v = (co < Co && ci < Ci) ? w[((long)co * Ci + ci) * kk + tap] : 0.f;
        wf[i] = (T)v;
        if (wd) wd[((long)ci * kk + tap) * CoPad + co] = (T)v;
    }
}

// Table-driven pack of every conv's fp32 OIHW master into the two MFMA operand layouts.  One block = one 64 (co) x 64 (ci)
// tile with all ks*ks taps: the master is read in contiguous runs (64 * ks*ks floats per output channel), transposed
// through LDS, and both packed matrices are written in 64-element runs.  (The previous element-per-thread gather read
// the master with a stride of ks*ks floats: rocprofv3 FETCH_SIZE showed 4.7 GB fetched for 256 MB of weights.)
// blocks[2b + 1] = tile index * 4 (the generic table counts CY_MULTI_ELEMS "elements" per block).
template <typename T, int KK>
__device__ __forceinline__ void pack_tile(const cy_pack_desc& d, int tile, T* lds) {
    constexpr int CH = Elem<T>::CH;          // elements per 16 bytes
    const int kk = KK > 0 ? KK : d.ks * d.ks;   // KK = 9 / 1: compile-time taps (the divisions below become multiplies)
    const int tiles_ci = (d.CiPad + 63) / 64;
    const int co0 = (tile / tiles_ci) * 64, ci0 = (tile % tiles_ci) * 64;
    const int ROW = kk * 64 + CH;            // elements; one 16-byte chunk of padding keeps rows aligned and staggers the banks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seg = 64 * kk;
    const bool vec = ((d.Ci * kk) & 3) == 0 && ci0 + 64 <= d.Ci;   // whole 16-byte groups of the master row are valid
    if (KK > 0 && vec) {
        // the tile's 64 rows of 64 * kk floats as one list of float4 groups, NB of them requested back to back per thread
        // before the first is scattered into LDS (one group per trip costs an HBM round trip per 16 bytes: 144 in a row for
        // a 3x3 tile, which is what the 0.29 ms of round 2's pack were made of)
        constexpr int KQ = KK > 0 ? KK : 1;
        constexpr int G4 = 16 * KQ;              // float4 groups per row
        constexpr int PER = 64 * G4 / 256;       // groups per thread: 4 * kk
        constexpr int NB = PER % 12 == 0 ? 12 : 4;
        static_assert(PER % NB == 0, "batches of loads");
#pragma unroll 1
        for (int b0 = 0; b0 < PER; b0 += NB) {
            typedef float fx4 __attribute__((ext_vector_type(4)));
            fx4 v[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int q = tid + 256 * (b0 + i), r = q / G4, j4 = (q - r * G4) * 4;
                // (row index clamped, value masked after the load: no branch between the loads; address space 1 because the
                // pointer comes out of the descriptor table and a flat load would also count against the LDS counter)
                const int rc = co0 + r < d.Co ? co0 + r : d.Co - 1;
                typedef __attribute__((address_space(1))) const fx4 gfx4;
                v[i] = *(gfx4*)(d.w + ((long)rc * d.Ci + ci0) * KQ + j4);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int q = tid + 256 * (b0 + i), r = q / G4, j4 = (q - r * G4) * 4;
                const float keep = co0 + r < d.Co ? 1.f : 0.f;
                const float f[4] = {v[i].x * keep, v[i].y * keep, v[i].z * keep, v[i].w * keep};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = j4 + e, ci_l = j / KQ, tap = j - ci_l * KQ;
                    lds[r * ROW + tap * 64 + ci_l] = (T)f[e];
                }
            }
        }
    } else {
        for (int r = wave; r < 64; r += 4) {
            const int co = co0 + r;
            const float* src = d.w + ((long)co * d.Ci + ci0) * kk;
            for (int j = lane; j < seg; j += 64) {
                const int ci_l = j / kk, tap = j - ci_l * kk;
                const float v = (co < d.Co && ci0 + ci_l < d.Ci) ? src[j] : 0.f;
                lds[r * ROW + tap * 64 + ci_l] = (T)v;
            }
        }
    }
    __syncthreads();
    T* wf = (T*)d.wf;
    T* wd = (T*)d.wd;
    constexpr int CPT = 64 / CH;             // 16-byte chunks per 64-element run
    const int nchunks = 64 * kk * CPT;
    for (int idx = tid; idx < nchunks; idx += 256) {         // forward layout [co][tap][ci]: 16 bytes of ci per thread
        const int cc = idx % CPT, t2 = idx / CPT;
        const int r = t2 / kk, tap = t2 - r * kk;
        const int co = co0 + r, ci = ci0 + cc * CH;
        if (co < d.CoPad && ci < d.CiPad)    // CiPad is a multiple of CH
            *reinterpret_cast<u32x4*>(wf + ((long)co * kk + tap) * d.CiPad + ci) =
                *reinterpret_cast<const u32x4*>(lds + r * ROW + tap * 64 + cc * CH);
    }
    if (wd) {
        for (int idx = tid; idx < nchunks; idx += 256) {     // dgrad layout [ci][tap][co]: 16 bytes of co per thread
            const int cc = idx % CPT, t2 = idx / CPT;
            const int ci_l = t2 / kk, tap = t2 - ci_l * kk;
            const int co = co0 + cc * CH, ci = ci0 + ci_l;
            if (co < d.CoPad && ci < d.CiPad) {
                T v[CH];
#pragma unroll
                for (int e = 0; e < CH; ++e) v[e] = lds[(cc * CH + e) * ROW + tap * 64 + ci_l];
                // (wd_ld: row stride of the dgrad matrix when this layer fills a column range of a WIDER one -- two sibling 1x1
                // layers packed side by side, [ci][Ca + Cb]; 0 = its own kk * CoPad)
                *reinterpret_cast<u32x4*>(wd + (d.wd_ld > 0 ? (long)ci * d.wd_ld + (long)tap * d.CoPad : ((long)ci * kk + tap) * d.CoPad) + co) =
                    *reinterpret_cast<const u32x4*>(v);
            }
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) pack_weights_multi_kernel(const cy_pack_desc* __restrict__ desc,
                                                                const int* __restrict__ blocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pk_smem[];
    T* lds = reinterpret_cast<T*>(pk_smem);
    const cy_pack_desc d = desc[blocks[2 * blockIdx.x]];
    const int tile = blocks[2 * blockIdx.x + 1] >> 2;
    if (d.ks == 3) pack_tile<T, 9>(d, tile, lds);
    else if (d.ks == 1) pack_tile<T, 1>(d, tile, lds);
    else pack_tile<T, 0>(d, tile, lds);
}

struct AdamGroups {
    float lr[8], wd[8];
};
__global__ void __launch_bounds__(256) adam_multi_kernel(const cy_adam_desc* __restrict__ desc, const int* __restrict__ blocks,
                                                        float beta1, float beta2, float eps, float bc1, float bc2,
                                                        int zero_grad, AdamGroups grp, const int* __restrict__ skip,
                                                        const int* __restrict__ step_in, int* __restrict__ step_out,
                                                        const float* __restrict__ grp_dev) {
    const bool skipped = skip && *skip;   // a non-finite gradient was found (cy_grad_nonfinite): the step is skipped as a whole
    if (step_in) {
        // step count on the device (cy_adam_multi_dev): a skipped step does not advance it, so the bias corrections of
        // the next step are those of t, not t + 1 -- without the host ever reading the flag.  step_out == nullptr: the
        // counter was already advanced for this step by cy_step_tick (the replayable form: same pointers every step).
        const int t = step_out ? *step_in + 1 : *step_in;
        if (step_out && blockIdx.x == 0 && threadIdx.x == 0) *step_out = skipped ? t - 1 : t;
        bc1 = (float)(1.0 - pow((double)beta1, (double)t));
        bc2 = (float)(1.0 - pow((double)beta2, (double)t));
    }
    if (skipped) return;
    const cy_adam_desc d = desc[blocks[2 * blockIdx.x]];
    const long first = (long)blocks[2 * blockIdx.x + 1] * 256;
    // learning rate / weight decay per group: by value, or from a device array [lr x 8, wd x 8] (a captured launch reads it anew)
    const float lr = grp_dev ? grp_dev[d.group & 7] : grp.lr[d.group & 7];
    const float wdecay = grp_dev ? grp_dev[8 + (d.group & 7)] : grp.wd[d.group & 7];
    const float step = lr / bc1, rs2 = rsqrtf(bc2);
#pragma unroll
    for (int it = 0; it < CY_MULTI_ELEMS / 256; ++it) {
        const long i = first + it * 256 + threadIdx.x;
        if (i >= d.n) break;
        float g = d.g[i];
        const float p = d.p[i];
        if (wdecay != 0.f) g += wdecay * p;
        const float m = beta1 * d.m[i] + (1.f - beta1) * g;
        const float v = beta2 * d.v[i] + (1.f - beta2) * g * g;
        d.m[i] = m;
        d.v[i] = v;
        d.p[i] = p - step * m / (sqrtf(v) * rs2 + eps);
        if (zero_grad) d.g[i] = 0.f;
    }
}

// torch.optim.SGD (dampening 0): g += wd * p;  buf = first step ? g : momentum * buf + g;  p -= lr * (nesterov ? g + momentum * buf : buf)
__global__ void __launch_bounds__(256) sgd_multi_kernel(const cy_adam_desc* __restrict__ desc, const int* __restrict__ blocks,
                                                       float momentum, int nesterov, int first_step, int zero_grad,
                                                       AdamGroups grp, const int* __restrict__ skip) {
    if (skip && *skip) return;
    const cy_adam_desc d = desc[blocks[2 * blockIdx.x]];
    const long first = (long)blocks[2 * blockIdx.x + 1] * 256;
    const float lr = grp.lr[d.group & 7], wdecay = grp.wd[d.group & 7];
#pragma unroll
    for (int it = 0; it < CY_MULTI_ELEMS / 256; ++it) {
        const long i = first + it * 256 + threadIdx.x;
        if (i >= d.n) break;
        float g = d.g[i];
        const float p = d.p[i];
        if (wdecay != 0.f) g += wdecay * p;
        float upd = g;
        if (momentum != 0.f) {
            const float buf = first_step ? g : momentum * d.m[i] + g;
            d.m[i] = buf;
            upd = nesterov ? g + momentum * buf : buf;
        }
        d.p[i] = p - lr * upd;
        if (zero_grad) d.g[i] = 0.f;
    }
}

// flag = 1 when any gradient element is inf / nan (dynamic loss scaling: such a step must not reach the optimizer)
__global__ void __launch_bounds__(256) grad_nonfinite_kernel(const float4* __restrict__ g, long n4, const float* __restrict__ tail,
                                                            int ntail, int* __restrict__ flag) {
    bool bad = false;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = g[i];
        // exponent all ones <=> inf or nan; OR of the four words' tests
        bad |= ((__float_as_uint(v.x) & 0x7F800000u) == 0x7F800000u) | ((__float_as_uint(v.y) & 0x7F800000u) == 0x7F800000u) |
               ((__float_as_uint(v.z) & 0x7F800000u) == 0x7F800000u) | ((__float_as_uint(v.w) & 0x7F800000u) == 0x7F800000u);
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) bad |= (__float_as_uint(tail[threadIdx.x]) & 0x7F800000u) == 0x7F800000u;
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// gbias[c] += scale * sum over the M pixels of d[p][c] (C <= 32 head channels).  Thread = (channel t & 31, row lane
// t >> 5): a wave reads two rows of C contiguous floats per load instruction (the first version ran one block per
// channel with a stride of C floats: C-fold read amplification, 52 us per head).  Per-thread sums in double, the eight
// row lanes are folded through LDS, one float atomic per channel and block.
__global__ void __launch_bounds__(256) bias_grad_kernel(const float* __restrict__ d, long M, int C, float scale,
                                                       const float* __restrict__ scale_dev, float* gbias) {
    __shared__ double red[8][32];
    if (scale_dev) scale *= *scale_dev;
    const int c = threadIdx.x & 31, lane_r = threadIdx.x >> 5;
    double s = 0.0;
    if (c < C)
        for (long p = (long)blockIdx.x * 8 + lane_r; p < M; p += (long)gridDim.x * 8) s += (double)d[p * C + c];
    red[lane_r][c] = s;
    __syncthreads();
    if (threadIdx.x < C) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += red[r][threadIdx.x];
        atomicAdd(gbias + threadIdx.x, scale * (float)t);
    }
}

// Deterministic mode, stage 1 of a two-stage fold: table [rows][W] -> out [rows_out][W], slice s = rows [s RS, (s+1) RS)
// summed in row order (double), the rows read are zeroed.  One thread per column: consecutive threads read consecutive
// floats.  (One wave per channel over up to 92 k rows, as the finalisers fold, was 56 us per layer.)
__global__ void __launch_bounds__(256) fold_rows_kernel(float* __restrict__ bins, int rows, int W, int RS, float* __restrict__ out) {
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= W) return;
    const int r0 = blockIdx.y * RS, r1 = min(rows, r0 + RS);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
        float* p = bins + (size_t)r * W + w;
        const float a = p[0], b = p[W], c = p[2 * (size_t)W], d = p[3 * (size_t)W];
        p[0] = 0.f; p[W] = 0.f; p[2 * (size_t)W] = 0.f; p[3 * (size_t)W] = 0.f;
        s0 += (double)a; s1 += (double)b; s2 += (double)c; s3 += (double)d;
    }
    for (; r < r1; ++r) {
        float* p = bins + (size_t)r * W + w;
        s0 += (double)*p;
        *p = 0.f;
    }
    out[(size_t)blockIdx.y * W + w] = (float)((s0 + s1) + (s2 + s3));
}

// deterministic head-bias gradient: stage 1 = per-block partial rows (no atomics), stage 2 = one block folds them in order
__global__ void __launch_bounds__(256) bias_grad_part_kernel(const float* __restrict__ d, long M, int C, float* __restrict__ part) {
    __shared__ double red[8][32];
    const int c = threadIdx.x & 31, lane_r = threadIdx.x >> 5;
    double s = 0.0;
    if (c < C)
        for (long p = (long)blockIdx.x * 8 + lane_r; p < M; p += (long)gridDim.x * 8) s += (double)d[p * C + c];
    red[lane_r][c] = s;
    __syncthreads();
    if (threadIdx.x < 32) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += red[r][threadIdx.x];
        part[blockIdx.x * 32 + threadIdx.x] = threadIdx.x < C ? (float)t : 0.f;
    }
}
__global__ void bias_grad_fold_kernel(const float* __restrict__ part, int nblocks, int C, float scale,
                                      const float* __restrict__ scale_dev, float* gbias) {
    if (scale_dev) scale *= *scale_dev;
    const int c = threadIdx.x;
    if (c >= C) return;
    double t = 0.0;
    for (int b = 0; b < nblocks; ++b) t += (double)part[b * 32 + c];
    gbias[c] += scale * (float)t;
}

inline int grid_for(long total) {
    long b = (total + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

inline bool rowmap_ok(int C, int ch) {
    const int cpr = C / ch;
    return C % ch == 0 && cpr >= 1 && cpr <= 256 && (256 % cpr) == 0;
}

inline long bn_blocks() {     // target number of pixel blocks of the fused BN passes (CY_BN_BLOCKS; measured on one box: 1024 -> 844, 4096 -> 846, 8192 / 16384 -> 848-849 images/s)
    static long n = 0;
    if (!n) { const char* e = getenv("CY_BN_BLOCKS"); n = e ? atol(e) : 8192; if (n < 64) n = 64; }
    return n;
}

inline int ppb_for(long M, int C, int ch) {
    // pixels per block: aim at ~2048 blocks, at least 4 passes per thread row
    const int rpp = 256 / (C / ch);
    long ppb = (M + 2047) / 2048;
    if (ppb < 4L * rpp) ppb = 4L * rpp;
    ppb = (ppb + rpp - 1) / rpp * rpp;
    return (int)ppb;
}

}  // namespace

#define CY_DT_SWITCH(dtype, M)                                                                       \
    if (dtype == CY_F16) { M(f16) } else if (dtype == CY_BF16) { M(bf16) } else if (dtype == CY_F32) { M(float) } \
    else return CY_ERR_ARG;

extern "C" int cy_bn_act_fwd(const void* x, int ldx, void* y, int ldy, const void* res, int ldres, int64_t M, int C,
                             const float* scale, const float* shift, int act, int dtype, cy_stream_t s) {
    CY_ENTER();
    const int ch = dtype == CY_F32 ? 4 : 8;
    if (!x || !y || !scale || !shift || !rowmap_ok(C, ch) || ldx % ch || ldy % ch || (res && ldres % ch)) return CY_ERR_ARG;
    const int ppb = ppb_for(M, C, ch);
    const dim3 grid((unsigned)((M + ppb - 1) / ppb));
#define CY_BNF(T, A)                                                                                               \
    if (res) hipLaunchKernelGGL((bn_act_fwd_kernel<T, A, true>), grid, dim3(256), 0, cy_s(s), (const T*)x, ldx, (T*)y, \
                                ldy, (const T*)res, ldres, (long)M, C, scale, shift, ppb);                         \
    else hipLaunchKernelGGL((bn_act_fwd_kernel<T, A, false>), grid, dim3(256), 0, cy_s(s), (const T*)x, ldx, (T*)y,   \
                            ldy, (const T*)nullptr, 0, (long)M, C, scale, shift, ppb);
#define CY_BNF_ACT(T)                                        \
    if (act == CY_ACT_MISH) { CY_BNF(T, CY_ACT_MISH) }       \
    else if (act == CY_ACT_LEAKY) { CY_BNF(T, CY_ACT_LEAKY) } \
    else { CY_BNF(T, CY_ACT_LINEAR) }
    CY_DT_SWITCH(dtype, CY_BNF_ACT)
#undef CY_BNF
#undef CY_BNF_ACT
    CY_LAUNCH_CHECK();
    return 0;
}

// channel group / pixels per block of the fused kernels: CG <= 128 channels with 256 / (CG / ch) rows per pass
static inline int fused_cg(int C, int ch) {
    // the largest CG = ch * 2^j <= 128 that divides C (2^j chunks per pixel row divide the block's 256 threads).  C % ch == 0 is
    // the callers' precondition, so CG = ch always qualifies.  (Round 4: the halving search this replaces walked 120 -> 60 -> 30
    // -> 15 for C = 120 and launched with CG = 15, not a multiple of the chunk: wrong outputs for channel counts that are not
    // 2^j * (a divisor of 128) -- none in the reference's cfgs, found by the ragged-channel case of the two-phase conv test.)
    for (int k = 128 / ch; k >= 1; k >>= 1)
        if (k * ch <= C && C % (k * ch) == 0) return k * ch;
    return 0;
}

extern "C" int cy_bn_act_fwd_fused(const void* x, int ldx, void* y, int ldy, const void* res, int ldres, int64_t M, int C,
                                   const float* stats_bins, int rows, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                                   float eps, float* vec_out, float* zero_table, int zero_n, int act, int dtype,
                                   int stats_ld, int stats_c0, int vec_ld, cy_stream_t s) {
    CY_ENTER();
    const int ch = dtype == CY_F32 ? 4 : 8;
    if (vec_ld != 0 && vec_ld < C) return CY_ERR_ARG;
    if (!x || !y || !stats_bins || !gamma || !beta || !vec_out || rows != CY_BINS || M < 1 || C % ch || ldx % ch || ldy % ch ||
        (res && ldres % ch) || (zero_n > 0 && !zero_table) || zero_table == stats_bins)
        return CY_ERR_ARG;
    const int cg = fused_cg(C, ch);
    if (cg < ch || cg > 128) return CY_ERR_ARG;
    const int rpp = 256 / (cg / ch);
    long ppb = (M + bn_blocks() - 1) / bn_blocks();                 // ~1024 pixel blocks x C / CG channel groups
    if (ppb < 4L * rpp) ppb = 4L * rpp;
    ppb = (ppb + rpp - 1) / rpp * rpp;
    const dim3 grid((unsigned)((M + ppb - 1) / ppb), (unsigned)(C / cg));
    BnFoldParams f;
    f.bins = stats_bins; f.zero_table = zero_table; f.zero_n = zero_n; f.count = (double)M; f.gamma = gamma; f.beta = beta;
    f.rmean = running_mean; f.rvar = running_var; f.nbt = (long long*)num_batches_tracked; f.momentum = momentum; f.eps = eps;
    f.vec = vec_out; f.vec_ld = vec_ld > 0 ? vec_ld : C;
    f.bins_ld = stats_ld > 0 ? stats_ld : C; f.bins_c0 = stats_ld > 0 ? stats_c0 : 0;
    if (f.bins_c0 < 0 || f.bins_c0 + C > f.bins_ld) return CY_ERR_ARG;
#define CY_BNFF(T, A)                                                                                                  \
    if (res) hipLaunchKernelGGL((bn_act_fwd_fused_kernel<T, A, true>), grid, dim3(256), 0, cy_s(s), (const T*)x, ldx, (T*)y, \
                                ldy, (const T*)res, ldres, (long)M, C, cg, f, (int)ppb);                                \
    else hipLaunchKernelGGL((bn_act_fwd_fused_kernel<T, A, false>), grid, dim3(256), 0, cy_s(s), (const T*)x, ldx, (T*)y,  \
                            ldy, (const T*)nullptr, 0, (long)M, C, cg, f, (int)ppb);
#define CY_BNFF_ACT(T)                                        \
    if (act == CY_ACT_MISH) { CY_BNFF(T, CY_ACT_MISH) }       \
    else if (act == CY_ACT_LEAKY) { CY_BNFF(T, CY_ACT_LEAKY) } \
    else { CY_BNFF(T, CY_ACT_LINEAR) }
    CY_DT_SWITCH(dtype, CY_BNFF_ACT)
#undef CY_BNFF
#undef CY_BNFF_ACT
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_bn_act_bwd_apply_fused(const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx, void* res_grad,
                                         int ldrg, int res_accum, int64_t M, int C, const float* mean, const float* invstd,
                                         const float* scale, const float* shift, const float* part_bins, int rows,
                                         float* ggamma, float* gbeta, float gscale, float* zero_table, int zero_n, int act,
                                         int dtype, int bins_ld, int bins_c0, cy_stream_t s) {
    CY_ENTER();
    const int ch = dtype == CY_F32 ? 4 : 8;
    if (!x || !dy || !dx || !part_bins || !mean || !invstd || !scale || !shift || rows != CY_BINS || M < 1 || C % ch || ldx % ch ||
        lddy % ch || lddx % ch || (res_grad && ldrg % ch) || (zero_n > 0 && !zero_table) || zero_table == part_bins)
        return CY_ERR_ARG;
    const int cg = fused_cg(C, ch);
    if (cg < ch || cg > 128) return CY_ERR_ARG;
    const int rpp = 256 / (cg / ch);
    long ppb = (M + bn_blocks() - 1) / bn_blocks();
    if (ppb < 4L * rpp) ppb = 4L * rpp;
    ppb = (ppb + rpp - 1) / rpp * rpp;
    const dim3 grid((unsigned)((M + ppb - 1) / ppb), (unsigned)(C / cg));
    BnBwdFoldParams f;
    f.bins = part_bins; f.zero_table = zero_table; f.zero_n = zero_n; f.ggamma = ggamma; f.gbeta = gbeta; f.gscale = gscale;
    f.bins_ld = bins_ld > 0 ? bins_ld : C; f.bins_c0 = bins_ld > 0 ? bins_c0 : 0;
    if (f.bins_c0 < 0 || f.bins_c0 + C > f.bins_ld) return CY_ERR_ARG;
#define CY_BNAF(T, A)                                                                                                 \
    hipLaunchKernelGGL((bn_bwd_apply_fused_kernel<T, A>), grid, dim3(256), 0, cy_s(s), (const T*)x, ldx, (const T*)dy, lddy, \
                       (T*)dx, lddx, (T*)res_grad, ldrg, res_accum, (long)M, C, cg, mean, invstd, scale, shift, f, (int)ppb);
#define CY_BNAF_ACT(T)                                        \
    if (act == CY_ACT_MISH) { CY_BNAF(T, CY_ACT_MISH) }       \
    else if (act == CY_ACT_LEAKY) { CY_BNAF(T, CY_ACT_LEAKY) } \
    else { CY_BNAF(T, CY_ACT_LINEAR) }
    CY_DT_SWITCH(dtype, CY_BNAF_ACT)
#undef CY_BNAF
#undef CY_BNAF_ACT
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_bn_scratch_rows(void) { return 0; }

extern "C" int cy_bn_bwd_rows(int64_t M, int C, int dtype) {
    CY_ENTER();
    const int ch = dtype == CY_F32 ? 4 : 8;
    if (!rowmap_ok(C, ch)) return CY_ERR_ARG;
    (void)M;
    return CY_BINS;
}

extern "C" int cy_bn_bwd_rows_det(int64_t M, int C, int dtype) {
    CY_ENTER();
    const int ch = dtype == CY_F32 ? 4 : 8;
    if (!rowmap_ok(C, ch)) return CY_ERR_ARG;
    const int ppb = ppb_for(M, C, ch);
    return (int)((M + ppb - 1) / ppb);
}

extern "C" int cy_bn_act_bwd_reduce(const void* x, int ldx, const void* dy, int lddy, int64_t M, int C,
                                    const float* mean, const float* invstd, const float* scale, const float* shift,
                                    int act, int dtype, float* part, int rows, cy_stream_t s) {
    CY_ENTER();
    const int ch = dtype == CY_F32 ? 4 : 8;
    if (!x || !dy || !part || !rowmap_ok(C, ch) || ldx % ch || lddy % ch) return CY_ERR_ARG;
    const int ppb = ppb_for(M, C, ch);
    const dim3 grid((unsigned)((M + ppb - 1) / ppb));
    if (rows != CY_BINS && rows != (int)grid.x) return CY_ERR_ARG;
    const int det = rows != CY_BINS;
#define CY_BNR(T, A)                                                                                            \
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, A>), grid, dim3(256), 0, cy_s(s), (const T*)x, ldx, (const T*)dy, \
                       lddy, (long)M, C, mean, invstd, scale, shift, ppb, part, det);
#define CY_BNR_ACT(T)                                        \
    if (act == CY_ACT_MISH) { CY_BNR(T, CY_ACT_MISH) }       \
    else if (act == CY_ACT_LEAKY) { CY_BNR(T, CY_ACT_LEAKY) } \
    else { CY_BNR(T, CY_ACT_LINEAR) }
    CY_DT_SWITCH(dtype, CY_BNR_ACT)
#undef CY_BNR
#undef CY_BNR_ACT
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_bn_act_bwd_apply(const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx,
                                   void* res_grad, int ldrg, int res_accum, int64_t M, int C, const float* mean,
                                   const float* invstd, const float* scale, const float* shift,
                                   const float* dgamma_sum, const float* dbeta_sum, int act, int dtype,
                                   cy_stream_t s) {
    CY_ENTER();
    const int ch = dtype == CY_F32 ? 4 : 8;
    if (!x || !dy || !dx || !rowmap_ok(C, ch) || ldx % ch || lddy % ch || lddx % ch || (res_grad && ldrg % ch))
        return CY_ERR_ARG;
    const int ppb = ppb_for(M, C, ch);
    const dim3 grid((unsigned)((M + ppb - 1) / ppb));
#define CY_BNA(T, A)                                                                                           \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, A>), grid, dim3(256), 0, cy_s(s), (const T*)x, ldx, (const T*)dy, \
                       lddy, (T*)dx, lddx, (T*)res_grad, ldrg, res_accum, (long)M, C, mean, invstd, scale, shift, \
                       dgamma_sum, dbeta_sum, ppb);
#define CY_BNA_ACT(T)                                        \
    if (act == CY_ACT_MISH) { CY_BNA(T, CY_ACT_MISH) }       \
    else if (act == CY_ACT_LEAKY) { CY_BNA(T, CY_ACT_LEAKY) } \
    else { CY_BNA(T, CY_ACT_LINEAR) }
    CY_DT_SWITCH(dtype, CY_BNA_ACT)
#undef CY_BNA
#undef CY_BNA_ACT
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_bn_finalize(const float* stats_part, int rows, int C, int64_t count, const float* gamma,
                              const float* beta, float* running_mean, float* running_var,
                              int64_t* num_batches_tracked, float momentum, float eps, float* mean, float* invstd,
                              float* scale, float* shift, cy_stream_t s) {
    CY_ENTER();
    if (!stats_part || !gamma || !beta || !mean || !invstd || !scale || !shift || rows < 1 || count < 1)
        return CY_ERR_ARG;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, cy_s(s), const_cast<float*>(stats_part), rows, C,
                       (double)count, gamma, beta, running_mean, running_var, (long long*)num_batches_tracked, momentum,
                       eps, mean, invstd, scale, shift);
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                                 const float* running_var, int C, float eps, float* scale, float* shift,
                                 cy_stream_t s) {
    CY_ENTER();
    if (!gamma || !beta || !running_mean || !running_var || !scale || !shift) return CY_ERR_ARG;
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, cy_s(s), gamma, beta, running_mean,
                       running_var, C, eps, scale, shift);
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_bn_bwd_finalize(const float* part, int rows, int C, float* dgamma_sum, float* dbeta_sum,
                                  float* ggamma, float* gbeta, float gscale, cy_stream_t s) {
    CY_ENTER();
    if (!part || !dgamma_sum || !dbeta_sum || rows < 1) return CY_ERR_ARG;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, cy_s(s), const_cast<float*>(part), rows, C,
                       dgamma_sum, dbeta_sum, ggamma, gbeta, gscale);
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_slice_copy(const void* x, int ldx, void* y, int ldy, int64_t M, int C, int accumulate, int dtype,
                             cy_stream_t s) {
    CY_ENTER();
    const int ch = dtype == CY_F32 ? 4 : 8;
    if (!x || !y || C % ch || ldx % ch || ldy % ch) return CY_ERR_ARG;
    const int g = grid_for(M * (C / ch));
#define CY_SL(T)                                                                                                     \
    if (accumulate) hipLaunchKernelGGL((slice_kernel<T, 1>), dim3(g), dim3(256), 0, cy_s(s), (const T*)x, ldx,       \
                                       (const T*)nullptr, 0, (T*)y, ldy, (long)M, C);                                \
    else hipLaunchKernelGGL((slice_kernel<T, 0>), dim3(g), dim3(256), 0, cy_s(s), (const T*)x, ldx, (const T*)nullptr, \
                            0, (T*)y, ldy, (long)M, C);
    CY_DT_SWITCH(dtype, CY_SL)
#undef CY_SL
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_slice_add(const void* a, int lda, const void* b, int ldb, void* y, int ldy, int64_t M, int C,
                            int dtype, cy_stream_t s) {
    CY_ENTER();
    const int ch = dtype == CY_F32 ? 4 : 8;
    if (!a || !b || !y || C % ch || lda % ch || ldb % ch || ldy % ch) return CY_ERR_ARG;
    const int g = grid_for(M * (C / ch));
#define CY_SA(T) \
    hipLaunchKernelGGL((slice_kernel<T, 2>), dim3(g), dim3(256), 0, cy_s(s), (const T*)a, lda, (const T*)b, ldb, (T*)y, \
                       ldy, (long)M, C);
    CY_DT_SWITCH(dtype, CY_SA)
#undef CY_SA
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t cy_maxpool_argmax_bytes(int N, int H, int OH, int OW, int C) { return (int64_t)N * (OH + H) * OW * C; }

extern "C" int cy_maxpool_fwd(const void* x, int N, int H, int W, int C, int ldx, void* y, int OH, int OW, int ldy,
                              int k, int stride, int pad, uint8_t* argmax, void* scratch, int dtype, cy_stream_t s) {
    CY_ENTER();
    const int ch = dtype == CY_F32 ? 4 : 8;
    if (!x || !y || !scratch || C % ch || ldx % ch || ldy % ch || k < 1 || k > 15 || stride < 1) return CY_ERR_ARG;
    uint8_t* a1 = argmax;
    uint8_t* b1 = argmax ? argmax + (size_t)N * OH * OW * C : nullptr;
    const int g1 = grid_for((long)N * H * OW * (C / ch));
    const int g2 = grid_for((long)N * OH * OW * (C / ch));
#define CY_MP(T)                                                                                                        \
    hipLaunchKernelGGL((maxpool_rows_kernel<T>), dim3(g1), dim3(256), 0, cy_s(s), (const T*)x, N, H, W, C, ldx, (T*)scratch, \
                       OW, k, stride, pad, b1);                                                                         \
    hipLaunchKernelGGL((maxpool_cols_kernel<T>), dim3(g2), dim3(256), 0, cy_s(s), (const T*)scratch, N, H, C, (T*)y, OH,  \
                       OW, ldy, k, stride, pad, a1);
    CY_DT_SWITCH(dtype, CY_MP)
#undef CY_MP
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_maxpool_bwd(const void* dy, int N, int OH, int OW, int C, int lddy, const uint8_t* argmax, void* dx,
                              int H, int W, int lddx, int k, int stride, int pad, int accumulate, float* scratch,
                              int dtype, cy_stream_t s) {
    CY_ENTER();
    const int ch = dtype == CY_F32 ? 4 : 8;
    if (!dy || !argmax || !dx || !scratch || C % ch || lddy % ch || lddx % ch || stride < 1) return CY_ERR_ARG;
    const uint8_t* a1 = argmax;
    const uint8_t* b1 = argmax + (size_t)N * OH * OW * C;
    const int g1 = grid_for((long)N * H * OW * (C / ch));
    const int g2 = grid_for((long)N * H * W * (C / ch));
#define CY_MB2(T, S1_)                                                                                                        \
    hipLaunchKernelGGL((maxpool_bwd_cols_kernel<T, S1_>), dim3(g1), dim3(256), 0, cy_s(s), (const T*)dy, N, OH, OW, C, lddy, a1,  \
                       scratch, H, k, stride, pad);                                                                            \
    hipLaunchKernelGGL((maxpool_bwd_rows_kernel<T, S1_>), dim3(g2), dim3(256), 0, cy_s(s), (const float*)scratch, b1, N, H, OW, C, \
                       (T*)dx, W, lddx, k, stride, pad, accumulate);
#define CY_MB(T) if (stride == 1) { CY_MB2(T, true) } else { CY_MB2(T, false) }
    CY_DT_SWITCH(dtype, CY_MB)
#undef CY_MB2
#undef CY_MB
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_upsample_fwd(const void* x, int N, int H, int W, int C, int ldx, void* y, int ldy, int stride,
                               int dtype, cy_stream_t s) {
    CY_ENTER();
    const int ch = dtype == CY_F32 ? 4 : 8;
    if (!x || !y || C % ch || ldx % ch || ldy % ch || stride < 1) return CY_ERR_ARG;
    const int g = grid_for((long)N * H * W * stride * stride * (C / ch));
#define CY_UF(T) \
    hipLaunchKernelGGL((upsample_fwd_kernel<T>), dim3(g), dim3(256), 0, cy_s(s), (const T*)x, N, H, W, C, ldx, (T*)y, ldy, stride);
    CY_DT_SWITCH(dtype, CY_UF)
#undef CY_UF
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_upsample_bwd(const void* dy, int N, int H, int W, int C, int lddy, void* dx, int lddx, int stride,
                               int accumulate, int dtype, cy_stream_t s) {
    CY_ENTER();
    const int ch = dtype == CY_F32 ? 4 : 8;
    if (!dy || !dx || C % ch || lddy % ch || lddx % ch || stride < 1) return CY_ERR_ARG;
    const int g = grid_for((long)N * H * W * (C / ch));
#define CY_UB(T) \
    hipLaunchKernelGGL((upsample_bwd_kernel<T>), dim3(g), dim3(256), 0, cy_s(s), (const T*)dy, N, H, W, C, lddy, (T*)dx, lddx, stride, accumulate);
    CY_DT_SWITCH(dtype, CY_UB)
#undef CY_UB
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_f32_to_view(const float* x, int64_t M, int C, float scale, const float* scale_dev, void* y,
                              int ldy, int CPad, int dtype, cy_stream_t s) {
    CY_ENTER();
    if (!x || !y || CPad < C || ldy < CPad) return CY_ERR_ARG;
    const int g = grid_for(M * CPad);
#define CY_FV(T) \
    hipLaunchKernelGGL((f32_to_view_kernel<T>), dim3(g), dim3(256), 0, cy_s(s), x, (long)M, C, scale, scale_dev, (T*)y, ldy, CPad, 0);
    CY_DT_SWITCH(dtype, CY_FV)
#undef CY_FV
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_nchw_to_nhwc(const float* x, int N, int C, int H, int W, int CPad, int dtype, void* out,
                               cy_stream_t s) {
    CY_ENTER();
    if (!x || !out || CPad < C) return CY_ERR_ARG;
    const int g = grid_for((long)N * H * W);
#define CY_NH(T) \
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), dim3(g), dim3(256), 0, cy_s(s), x, N, C, H, W, CPad, (T*)out);
    CY_DT_SWITCH(dtype, CY_NH)
#undef CY_NH
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_pack_weights(const float* w, int Co, int Ci, int ks, int CoPad, int CiPad, int dtype, void* wf,
                               void* wd, cy_stream_t s) {
    CY_ENTER();
    if (!w || !wf || CoPad < Co || CiPad < Ci) return CY_ERR_ARG;
    const int g = grid_for((long)CoPad * ks * ks * CiPad);
#define CY_PW(T) \
    hipLaunchKernelGGL((pack_weights_kernel<T>), dim3(g), dim3(256), 0, cy_s(s), w, Co, Ci, ks, CoPad, CiPad, (T*)wf, (T*)wd);
    CY_DT_SWITCH(dtype, CY_PW)
#undef CY_PW
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_pack_weights_multi(const cy_pack_desc* desc, const int32_t* blocks, int nblocks, int dtype,
                                     cy_stream_t s) {
    CY_ENTER();
    if (!desc || !blocks || nblocks < 1) return CY_ERR_ARG;
    // LDS for the largest tile (3x3 taps): 64 rows x (9 * 64 + one 16-byte chunk) elements (75 KB f16 / 148 KB f32)
    static unsigned long long attr_done = 0;      // bit d: set for HIP device d
    if (cy_first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pack_weights_multi_kernel<f16>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 64 * (9 * 64 + 8) * 2);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pack_weights_multi_kernel<bf16>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 64 * (9 * 64 + 8) * 2);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pack_weights_multi_kernel<float>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 64 * (9 * 64 + 4) * 4);
    }
    if (dtype == CY_F16)
        hipLaunchKernelGGL((pack_weights_multi_kernel<f16>), dim3(nblocks), dim3(256), 64 * (9 * 64 + 8) * 2, cy_s(s), desc, blocks);
    else if (dtype == CY_BF16)
        hipLaunchKernelGGL((pack_weights_multi_kernel<bf16>), dim3(nblocks), dim3(256), 64 * (9 * 64 + 8) * 2, cy_s(s), desc, blocks);
    else if (dtype == CY_F32)
        hipLaunchKernelGGL((pack_weights_multi_kernel<float>), dim3(nblocks), dim3(256), 64 * (9 * 64 + 4) * 4, cy_s(s), desc, blocks);
    else
        return CY_ERR_ARG;
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_adam_multi(const cy_adam_desc* desc, const int32_t* blocks, int nblocks, float beta1, float beta2,
                             float eps, float bias_corr1, float bias_corr2, int zero_grad, const float* group_lr_host,
                             const float* group_wd_host, int ngroups, const int32_t* skip_flag, cy_stream_t s) {
    CY_ENTER();
    if (!desc || !blocks || nblocks < 1 || bias_corr1 <= 0.f || bias_corr2 <= 0.f) return CY_ERR_ARG;
    if (!group_lr_host || !group_wd_host || ngroups < 1 || ngroups > 8) return CY_ERR_ARG;
    AdamGroups grp;
    for (int i = 0; i < 8; ++i) {
        grp.lr[i] = i < ngroups ? group_lr_host[i] : 0.f;
        grp.wd[i] = i < ngroups ? group_wd_host[i] : 0.f;
    }
    hipLaunchKernelGGL(adam_multi_kernel, dim3(nblocks), dim3(256), 0, cy_s(s), desc, blocks, beta1, beta2, eps, bias_corr1,
                       bias_corr2, zero_grad, grp, (const int*)skip_flag, (const int*)nullptr, (int*)nullptr, (const float*)nullptr);
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_adam_multi_dev(const cy_adam_desc* desc, const int32_t* blocks, int nblocks, float beta1, float beta2,
                                 float eps, const int32_t* step_in, int32_t* step_out, int zero_grad,
                                 const float* group_lr_host, const float* group_wd_host, int ngroups,
                                 const int32_t* skip_flag, cy_stream_t s) {
    CY_ENTER();
    if (!desc || !blocks || nblocks < 1 || !step_in || !step_out || step_in == step_out) return CY_ERR_ARG;
    if (!group_lr_host || !group_wd_host || ngroups < 1 || ngroups > 8) return CY_ERR_ARG;
    AdamGroups grp;
    for (int i = 0; i < 8; ++i) {
        grp.lr[i] = i < ngroups ? group_lr_host[i] : 0.f;
        grp.wd[i] = i < ngroups ? group_wd_host[i] : 0.f;
    }
    hipLaunchKernelGGL(adam_multi_kernel, dim3(nblocks), dim3(256), 0, cy_s(s), desc, blocks, beta1, beta2, eps, 1.f, 1.f,
                       zero_grad, grp, (const int*)skip_flag, (const int*)step_in, (int*)step_out, (const float*)nullptr);
    CY_LAUNCH_CHECK();
    return 0;
}

__global__ void step_tick_kernel(int* __restrict__ counter, const int* __restrict__ skip) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && !(skip && *skip)) *counter += 1;
}

// The replayable (hipGraph-capturable) optimizer step: nothing that changes from step to step is passed by value.
extern "C" int cy_adam_multi_graph(const cy_adam_desc* desc, const int32_t* blocks, int nblocks, float beta1, float beta2,
                                   float eps, int32_t* step_counter, const float* group_lr_wd_dev, int zero_grad,
                                   const int32_t* skip_flag, cy_stream_t s) {
    CY_ENTER();
    if (!desc || !blocks || nblocks < 1 || !step_counter || !group_lr_wd_dev) return CY_ERR_ARG;
    AdamGroups grp;
    for (int i = 0; i < 8; ++i) { grp.lr[i] = 0.f; grp.wd[i] = 0.f; }
    hipLaunchKernelGGL(step_tick_kernel, dim3(1), dim3(64), 0, cy_s(s), (int*)step_counter, (const int*)skip_flag);
    hipLaunchKernelGGL(adam_multi_kernel, dim3(nblocks), dim3(256), 0, cy_s(s), desc, blocks, beta1, beta2, eps, 1.f, 1.f,
                       zero_grad, grp, (const int*)skip_flag, (const int*)step_counter, (int*)nullptr, group_lr_wd_dev);
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_sgd_multi(const cy_adam_desc* desc, const int32_t* blocks, int nblocks, float momentum, int nesterov,
                            int first_step, int zero_grad, const float* group_lr_host, const float* group_wd_host, int ngroups,
                            const int32_t* skip_flag, cy_stream_t s) {
    CY_ENTER();
    if (!desc || !blocks || nblocks < 1 || momentum < 0.f) return CY_ERR_ARG;
    if (!group_lr_host || !group_wd_host || ngroups < 1 || ngroups > 8) return CY_ERR_ARG;
    AdamGroups grp;
    for (int i = 0; i < 8; ++i) {
        grp.lr[i] = i < ngroups ? group_lr_host[i] : 0.f;
        grp.wd[i] = i < ngroups ? group_wd_host[i] : 0.f;
    }
    hipLaunchKernelGGL(sgd_multi_kernel, dim3(nblocks), dim3(256), 0, cy_s(s), desc, blocks, momentum, nesterov, first_step,
                       zero_grad, grp, (const int*)skip_flag);
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_grad_nonfinite(const float* g, int64_t n, int32_t* flag, cy_stream_t s) {
    CY_ENTER();
    if (!g || !flag || n < 0 || ((uintptr_t)g & 15)) return CY_ERR_ARG;
    if (hipMemsetAsync(flag, 0, sizeof(int32_t), cy_s(s)) != hipSuccess) return -(1000 + 1);
    const long n4 = n / 4;
    const int ntail = (int)(n - n4 * 4);
    hipLaunchKernelGGL(grad_nonfinite_kernel, dim3(grid_for(n4 > 0 ? n4 : 1)), dim3(256), 0, cy_s(s),
                       reinterpret_cast<const float4*>(g), n4, g + n4 * 4, ntail, (int*)flag);
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_bias_grad(const float* dlogits, int64_t M, int C, float scale, const float* scale_dev,
                            float* gbias, int deterministic, cy_stream_t s) {
    CY_ENTER();
    if (!dlogits || !gbias || C < 1 || C > 32) return CY_ERR_ARG;
    long blocks = (M + 8 * 32 - 1) / (8 * 32);   // >= 32 rows per row lane
    if (blocks > 512) blocks = 512;
    if (blocks < 1 || deterministic) blocks = 1;  // one block = one add per channel: the sum does not depend on block order
    hipLaunchKernelGGL(bias_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, cy_s(s), dlogits, (long)M, C, scale, scale_dev, gbias);
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_fold_rows(float* bins, int rows, int W, float* out, int rows_out, cy_stream_t s) {
    CY_ENTER();
    if (!bins || !out || rows < 1 || W < 1 || rows_out < 1 || rows_out > rows) return CY_ERR_ARG;
    const int RS = (rows + rows_out - 1) / rows_out;
    if ((rows + RS - 1) / RS != rows_out) return CY_ERR_ARG;      // rows_out must be cy_fold_rows_out(rows)
    hipLaunchKernelGGL(fold_rows_kernel, dim3((W + 255) / 256, rows_out), dim3(256), 0, cy_s(s), bins, rows, W, RS, out);
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_fold_rows_out(int rows) {
    CY_ENTER();
    if (rows <= 256) return rows;
    const int RS = (rows + 127) / 128;
    return (rows + RS - 1) / RS;
}

extern "C" int cy_bias_grad_det(const float* dlogits, int64_t M, int C, float scale, const float* scale_dev, float* gbias,
                                float* scratch, cy_stream_t s) {
    CY_ENTER();
    if (!dlogits || !gbias || !scratch || C < 1 || C > 32 || M < 1) return CY_ERR_ARG;
    long blocks = (M + 8 * 32 - 1) / (8 * 32);
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(bias_grad_part_kernel, dim3((unsigned)blocks), dim3(256), 0, cy_s(s), dlogits, (long)M, C, scratch);
    hipLaunchKernelGGL(bias_grad_fold_kernel, dim3(1), dim3(32), 0, cy_s(s), scratch, (int)blocks, C, scale, scale_dev, gbias);
    CY_LAUNCH_CHECK();
    return 0;
}
