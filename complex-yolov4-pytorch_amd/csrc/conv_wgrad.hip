// Weight gradient on gfx950 MFMA:  dW[co][(tap,ci)] = sum_pixels dY[pixel][co] * X[pixel (+) tap][ci].
//
// The reduction index is the pixel, while both operands are stored channel-contiguous (NHWC), so each
// MFMA fragment needs a transposed view of an LDS tile stored [pixel][channel].  f16 uses the gfx950
// LDS transpose read ds_read_b64_tr_b16 (two per 8-element fragment); a scalar-gather path
// (use_tr = 0) exists to pin that mapping in the test-suite.  f32 parity mode needs no transpose
// (one element per lane).  LDS rows are padded by 32 bytes: 8 consecutive pixel rows then start on
// distinct 32-byte bank groups, which is the access pattern of one transpose read.
//
// Grid = (column tiles of ks*ks*Ci, channel tiles of Co, split): split-K over pixels; each block
// writes its own slab part[sp] (deterministic; cy_wgrad_reduce folds the slabs).
#include <stdlib.h>

#include "common.hpp"

namespace {

struct WgradParams {
    const unsigned char* dy;
    const unsigned char* x;
    float* part;
    int N, OH, OW, Co, lddy;
    int XH, XW, Ci, ldx;
    int ks, stride, pad;
    int M, Ncols, CoRows;
    int pps;  // pixels per split (multiple of BKP)
    unsigned x_bytes;  // extent of the X view (wgrad_dma_kernel's descriptor); 0 = offsets do not fit 32 bits
    int ncol_tiles, nco_tiles;
    int atomic;        // 1: every split adds into slab 0 with fp32 atomics (one resident, pre-zeroed slab per layer) instead of
                       // storing its own slab -- the split-K slabs' write + fold traffic goes away; order-dependent rounding
};

// one accumulator tile -> its slab (plain stores) or slab 0 (fp32 atomics, fire-and-forget)
template <int TI, int TJ, int BCO, int BCI>
__device__ __forceinline__ void wgrad_store(const WgradParams& p, const f32x4 (&acc)[TI][TJ], int sp, int co0, int col0, int wi, int wj,
                                            int q16, int g) {
    float* slab = p.part + (p.atomic ? (size_t)0 : (size_t)sp * p.CoRows * p.Ncols);
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int col = col0 + wj * (BCI / 2) + j * 16 + q16;
            if (col >= p.Ncols) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wi * (BCO / 2) + i * 16 + g * 4 + r;
                if (co >= p.CoRows) continue;
                float* dst = slab + (size_t)co * p.Ncols + col;
                if (p.atomic) __hip_atomic_fetch_add(dst, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *dst = acc[i][j][r];
            }
        }
}

typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

template <typename T>
struct WTraits;
template <>
struct WTraits<f16> {
    static constexpr int BKP = 64;
};
template <>
struct WTraits<bf16> {
    static constexpr int BKP = 64;
};
template <>
struct WTraits<float> {
    static constexpr int BKP = 32;
};

// 16-bit MFMA on 8-element fragments of either storage type; the transpose read is type-blind (16-bit elements)
template <typename T>
struct WMma;
template <>
struct WMma<f16> {
    typedef f16x8 frag;
    __device__ static __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <>
struct WMma<bf16> {
    typedef bf16x8 frag;
    __device__ static __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <typename T>
__device__ __forceinline__ typename WMma<T>::frag tr16_pair(const unsigned char* lo_ptr, const unsigned char* hi_ptr) {
    const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(lo_ptr));
    const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(hi_ptr));
    u32x4 v;
    const u32x2 l = __builtin_bit_cast(u32x2, lo), h = __builtin_bit_cast(u32x2, hi);
    v[0] = l[0]; v[1] = l[1]; v[2] = h[0]; v[3] = h[1];
    return __builtin_bit_cast(typename WMma<T>::frag, v);
}

template <typename T, int BCO, int BCI, bool USE_TR>
__global__ void __launch_bounds__(256) wgrad_kernel(const WgradParams p) {
    constexpr int CH = Elem<T>::CH;
    constexpr int BKP = WTraits<T>::BKP;
    constexpr int CPR_A = BCO / CH, RPP_A = 256 / CPR_A, PASS_A = BKP / RPP_A;
    constexpr int CPR_B = BCI / CH, RPP_B = 256 / CPR_B, PASS_B = BKP / RPP_B;
    constexpr int RB_A = BCO * (int)sizeof(T) + 32, RB_B = BCI * (int)sizeof(T) + 32;
    constexpr int STAGE = BKP * (RB_A + RB_B);
    constexpr int TI = BCO / 32, TJ = BCI / 32;
    static_assert(PASS_A >= 1 && PASS_B >= 1, "tile too narrow");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1;
    // XCD-aware order (hardware places block b on XCD b % 8): all (column tile, channel tile) blocks of one pixel range
    // become neighbours on ONE XCD, so its L2 serves the dY / X tiles they share (PMC: 3.5x the algorithmic HBM fetch
    // without this).
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, slot = bid >> 3;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int ct = lid % p.ncol_tiles, rest = lid / p.ncol_tiles;
    const int col0 = ct * BCI, co0 = (rest % p.nco_tiles) * BCO, sp = rest / p.nco_tiles;
    const int pix_begin = sp * p.pps;
    const int pix_end = min(p.M, pix_begin + p.pps);

    // ---- staging coordinates -------------------------------------------------------------------
    const int a_chunk = tid % CPR_A, a_row0 = tid / CPR_A;
    const int b_chunk = tid % CPR_B, b_row0 = tid / CPR_B;
    const int a_co = co0 + a_chunk * CH;
    const bool a_cok = a_co < p.Co;  // Co is a multiple of CH for every caller
    const int b_col = col0 + b_chunk * CH;
    const bool b_cok = b_col < p.Ncols;
    const int b_tap = b_cok ? b_col / p.Ci : 0;
    const int b_ci = b_col - b_tap * p.Ci;
    const int b_kh = b_tap / p.ks, b_kw = b_tap - b_kh * p.ks;
    // pixel decomposition of this thread's B rows, advanced incrementally by BKP per K step
    int bn[PASS_B], boh[PASS_B], bow[PASS_B];
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int q = 0; q < PASS_B; ++q) {
        const int m = pix_begin + b_row0 + q * RPP_B;
        const int n = m / ohw, rem = m - n * ohw;
        bn[q] = n; boh[q] = rem / p.OW; bow[q] = rem - boh[q] * p.OW;
    }

    u32x4 av[PASS_A], bv[PASS_B];
    auto load_tile = [&](int pix0) {
#pragma unroll
        for (int q = 0; q < PASS_A; ++q) {
            const int m = pix0 + a_row0 + q * RPP_A;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (a_cok && m < pix_end) v = *reinterpret_cast<const u32x4*>(p.dy + ((size_t)m * p.lddy + a_co) * sizeof(T));
            av[q] = v;
        }
#pragma unroll
        for (int q = 0; q < PASS_B; ++q) {
            const int m = pix0 + b_row0 + q * RPP_B;
            u32x4 v = {0u, 0u, 0u, 0u};
            const int xh = boh[q] * p.stride - p.pad + b_kh, xw = bow[q] * p.stride - p.pad + b_kw;
            if (b_cok && m < pix_end && xh >= 0 && xh < p.XH && xw >= 0 && xw < p.XW) {
                const size_t off = ((size_t)((bn[q] * p.XH + xh) * p.XW + xw) * p.ldx + b_ci) * sizeof(T);
                v = *reinterpret_cast<const u32x4*>(p.x + off);
            }
            bv[q] = v;
            bow[q] += BKP;
            while (bow[q] >= p.OW) { bow[q] -= p.OW; ++boh[q]; }
            while (boh[q] >= p.OH) { boh[q] -= p.OH; ++bn[q]; }
        }
    };
    auto store_tile = [&](int stage) {
        unsigned char* as = smem + stage * STAGE;
        unsigned char* bs = as + BKP * RB_A;
#pragma unroll
        for (int q = 0; q < PASS_A; ++q)
            *reinterpret_cast<u32x4*>(as + (a_row0 + q * RPP_A) * RB_A + a_chunk * 16) = av[q];
#pragma unroll
        for (int q = 0; q < PASS_B; ++q)
            *reinterpret_cast<u32x4*>(bs + (b_row0 + q * RPP_B) * RB_B + b_chunk * 16) = bv[q];
    };

    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkt = pix_end > pix_begin ? (pix_end - pix_begin + BKP - 1) / BKP : 0;
    if (nkt > 0) {
        load_tile(pix_begin);
        store_tile(0);
    }
    __syncthreads();
    const int q16 = lane & 15, g = lane >> 4;
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) load_tile(pix_begin + (kt + 1) * BKP);
        const unsigned char* as = smem + cur * STAGE;
        const unsigned char* bs = as + BKP * RB_A;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int kk = 0; kk < BKP / 32; ++kk) {
                typename WMma<T>::frag a[TI], b[TJ];
                if constexpr (USE_TR) {
                    // lane supplies row (q16>>2) and 4 channels at (q16&3)*4 of a [4 pixel][16 channel] block;
                    // it receives channel q16 of the 4 pixel rows (hardware transpose within 16 lanes).
                    const int prow = kk * 32 + g * 8 + (q16 >> 2);
#pragma unroll
                    for (int i = 0; i < TI; ++i) {
                        const unsigned char* ptr = as + prow * RB_A + ((wi * (BCO / 2) + i * 16 + ((q16 & 3) << 2)) << 1);
                        a[i] = tr16_pair<T>(ptr, ptr + 4 * RB_A);
                    }
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
                        const unsigned char* ptr = bs + prow * RB_B + ((wj * (BCI / 2) + j * 16 + ((q16 & 3) << 2)) << 1);
                        b[j] = tr16_pair<T>(ptr, ptr + 4 * RB_B);
                    }
                } else {
                    const int prow = kk * 32 + g * 8;
#pragma unroll
                    for (int i = 0; i < TI; ++i)
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            a[i][e] = *reinterpret_cast<const T*>(as + (prow + e) * RB_A + ((wi * (BCO / 2) + i * 16 + q16) << 1));
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            b[j][e] = *reinterpret_cast<const T*>(bs + (prow + e) * RB_B + ((wj * (BCI / 2) + j * 16 + q16) << 1));
                }
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = WMma<T>::mma(a[i], b[j], acc[i][j]);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BKP / 4; ++kk) {
                float a[TI], b[TJ];
                const int prow = kk * 4 + g;
#pragma unroll
                for (int i = 0; i < TI; ++i)
                    a[i] = *reinterpret_cast<const float*>(as + prow * RB_A + ((wi * (BCO / 2) + i * 16 + q16) << 2));
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    b[j] = *reinterpret_cast<const float*>(bs + prow * RB_B + ((wj * (BCI / 2) + j * 16 + q16) << 2));
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        if (kt + 1 < nkt) store_tile(cur ^ 1);
        __syncthreads();
    }

    // D[co = ... + g*4 + r][col = ... + q16]
    wgrad_store<TI, TJ, BCO, BCI>(p, acc, sp, co0, col0, wi, wj, q16, g);
}

// ---------------------------------------------------------------------------------------------------------------------
// BatchNorm backward + weight gradient of a conv WITHOUT an input gradient (the first layer) in ONE kernel.
// That layer's pre-BN gradient dRaw has a single reader, its own weight gradient -- and at 608 x 608 x 32 channels it is the
// largest tensor of the step (378 MB at batch 16): the separate passes write it (cy_bn_act_bwd_apply_fused: read g + raw, write
// dRaw) and read it back (cy_conv_wgrad), the last two kernels of backward, on the critical path in front of the optimizer
// (profiles/r06_timeline: 238 + 158 us).  Here the register-staged kernel above applies
//     dRaw = scale * (g * act'(raw * scale + shift) - mean(dz) - xhat * mean(dz * xhat))
// to the (g, raw) chunks it has just loaded, rounds to the storage type exactly as the apply pass does, and writes the result
// into its LDS tile instead of memory: dRaw never exists.  Prologue as cy_bn_act_bwd_apply_fused: fold of the CY_STAT_BINS sum
// bins (double, bin order), the BatchNorm parameter gradients (first block), zeroing of the other table of the pair.
struct WgradBnParams {
    const unsigned char* raw;     // the layer's pre-BN tensor, same pixels as dy
    int ldraw;
    const float* mean;
    const float* invstd;
    const float* scale;
    const float* shift;
    const float* bins;            // [CY_STAT_BINS][2][Co]: sum dz, sum dz * xhat
    float* ggamma;
    float* gbeta;
    float gscale;
    float* zero_table;
    int zero_n;
};

template <typename T, int ACT>
__global__ void __launch_bounds__(256) wgrad_bn_kernel(const WgradParams p, const WgradBnParams f) {
    constexpr int CH = 8, BKP = 64, BCO = 32, BCI = 128;
    constexpr int CPR_A = BCO / CH, RPP_A = 256 / CPR_A, PASS_A = BKP / RPP_A;      // 4 chunks per row, 64 rows per pass, 1 pass
    constexpr int CPR_B = BCI / CH, RPP_B = 256 / CPR_B, PASS_B = BKP / RPP_B;
    constexpr int RB_A = BCO * 2 + 32, RB_B = BCI * 2 + 32;
    constexpr int STAGE = BKP * (RB_A + RB_B);
    constexpr int TI = BCO / 32, TJ = BCI / 32;
    static_assert(PASS_A == 1, "one dy / raw chunk per thread and K step");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float s_dg[BCO], s_db[BCO];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1;
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, slot = bid >> 3;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int ct = lid % p.ncol_tiles, sp = lid / p.ncol_tiles;      // one tile of output channels (Co <= 32)
    const int col0 = ct * BCI, co0 = 0;
    const int pix_begin = sp * p.pps;
    const int pix_end = min(p.M, pix_begin + p.pps);

    // ---- BatchNorm-backward fold (every block needs the two means; the first one also owns the parameter gradients) ----
    if (tid < BCO) {
        double s1 = 0.0, s2 = 0.0;
        if (tid < p.Co) {
#pragma unroll
            for (int b = 0; b < CY_STAT_BINS; ++b) {
                s1 += (double)f.bins[((size_t)b * 2) * p.Co + tid];
                s2 += (double)f.bins[((size_t)b * 2 + 1) * p.Co + tid];
            }
            if (lid == 0) {
                if (f.gbeta) f.gbeta[tid] += f.gscale * (float)s1;
                if (f.ggamma) f.ggamma[tid] += f.gscale * (float)s2;
            }
        }
        s_db[tid] = (float)s1;
        s_dg[tid] = (float)s2;
    }
    for (int i = blockIdx.x * 256 + tid; i < f.zero_n; i += gridDim.x * 256) f.zero_table[i] = 0.f;
    __syncthreads();

    const int a_chunk = tid % CPR_A, a_row0 = tid / CPR_A;
    const int b_chunk = tid % CPR_B, b_row0 = tid / CPR_B;
    const int a_co = co0 + a_chunk * CH;
    const bool a_cok = a_co < p.Co;        // Co is a multiple of CH
    float sc[CH], sh[CH], mu[CH], is[CH], mg[CH], mb[CH];
    {
        const float invM = 1.f / (float)p.M;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = min(a_co + i, p.Co - 1);
            sc[i] = f.scale[c]; sh[i] = f.shift[c]; mu[i] = f.mean[c]; is[i] = f.invstd[c];
            mg[i] = s_dg[min(a_chunk * CH + i, BCO - 1)] * invM; mb[i] = s_db[min(a_chunk * CH + i, BCO - 1)] * invM;
        }
    }
    const int b_col = col0 + b_chunk * CH;
    const bool b_cok = b_col < p.Ncols;
    const int b_tap = b_cok ? b_col / p.Ci : 0;
    const int b_ci = b_col - b_tap * p.Ci;
    const int b_kh = b_tap / p.ks, b_kw = b_tap - b_kh * p.ks;
    int bn[PASS_B], boh[PASS_B], bow[PASS_B];
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int q = 0; q < PASS_B; ++q) {
        const int m = pix_begin + b_row0 + q * RPP_B;
        const int n = m / ohw, rem = m - n * ohw;
        bn[q] = n; boh[q] = rem / p.OW; bow[q] = rem - boh[q] * p.OW;
    }

    u32x4 gv, rv, bv[PASS_B];
    bool a_ok = false;
    auto load_tile = [&](int pix0) {
        const int m = pix0 + a_row0;
        a_ok = a_cok && m < pix_end;
        gv = u32x4{0u, 0u, 0u, 0u};
        rv = gv;
        if (a_ok) {
            gv = *reinterpret_cast<const u32x4*>(p.dy + ((size_t)m * p.lddy + a_co) * sizeof(T));
            rv = *reinterpret_cast<const u32x4*>(f.raw + ((size_t)m * f.ldraw + a_co) * sizeof(T));
        }
#pragma unroll
        for (int q = 0; q < PASS_B; ++q) {
            const int mq = pix0 + b_row0 + q * RPP_B;
            u32x4 v = {0u, 0u, 0u, 0u};
            const int xh = boh[q] * p.stride - p.pad + b_kh, xw = bow[q] * p.stride - p.pad + b_kw;
            if (b_cok && mq < pix_end && xh >= 0 && xh < p.XH && xw >= 0 && xw < p.XW) {
                const size_t off = ((size_t)((bn[q] * p.XH + xh) * p.XW + xw) * p.ldx + b_ci) * sizeof(T);
                v = *reinterpret_cast<const u32x4*>(p.x + off);
            }
            bv[q] = v;
            bow[q] += BKP;
            while (bow[q] >= p.OW) { bow[q] -= p.OW; ++boh[q]; }
            while (boh[q] >= p.OH) { boh[q] -= p.OH; ++bn[q]; }
        }
    };
    auto store_tile = [&](int stage) {
        unsigned char* as = smem + stage * STAGE;
        unsigned char* bs = as + BKP * RB_A;
        // the BatchNorm backward of this thread's 8 channels of one pixel (cy_bn_act_bwd_apply_fused's arithmetic, rounded to
        // the storage type like the tensor it replaces); rows outside the split contribute zeros
        u32x4 dv = {0u, 0u, 0u, 0u};
        if (a_ok) {
            float fv[CH], g[CH], o[CH];
            chunk_to_f32<T>(rv, fv);
            chunk_to_f32<T>(gv, g);
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const float dz = g[i] * act_grad<ACT, true>(fv[i] * sc[i] + sh[i]);
                const float xh = (fv[i] - mu[i]) * is[i];
                o[i] = sc[i] * (dz - mb[i] - xh * mg[i]);
            }
            dv = f32_to_chunk<T>(o);
        }
        *reinterpret_cast<u32x4*>(as + a_row0 * RB_A + a_chunk * 16) = dv;
#pragma unroll
        for (int q = 0; q < PASS_B; ++q)
            *reinterpret_cast<u32x4*>(bs + (b_row0 + q * RPP_B) * RB_B + b_chunk * 16) = bv[q];
    };

    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkt = pix_end > pix_begin ? (pix_end - pix_begin + BKP - 1) / BKP : 0;
    if (nkt > 0) {
        load_tile(pix_begin);
        store_tile(0);
    }
    __syncthreads();
    const int q16 = lane & 15, g = lane >> 4;
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) load_tile(pix_begin + (kt + 1) * BKP);
        const unsigned char* as = smem + cur * STAGE;
        const unsigned char* bs = as + BKP * RB_A;
#pragma unroll
        for (int kk = 0; kk < BKP / 32; ++kk) {
            typename WMma<T>::frag a[TI], b[TJ];
            const int prow = kk * 32 + g * 8 + (q16 >> 2);
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const unsigned char* ptr = as + prow * RB_A + ((wi * (BCO / 2) + i * 16 + ((q16 & 3) << 2)) << 1);
                a[i] = tr16_pair<T>(ptr, ptr + 4 * RB_A);
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const unsigned char* ptr = bs + prow * RB_B + ((wj * (BCI / 2) + j * 16 + ((q16 & 3) << 2)) << 1);
                b[j] = tr16_pair<T>(ptr, ptr + 4 * RB_B);
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = WMma<T>::mma(a[i], b[j], acc[i][j]);
        }
        if (kt + 1 < nkt) store_tile(cur ^ 1);
        __syncthreads();
    }
    wgrad_store<TI, TJ, BCO, BCI>(p, acc, sp, co0, col0, wi, wj, q16, g);
}

// ---------------------------------------------------------------------------------------------------------------------
// f16 weight gradient with direct-to-LDS tiles (the default f16 path; the register-staged kernel above stays for f32,
// for the scalar-gather check of the transpose-read mapping and for offsets beyond 32 bits).
//
// PMC of the register-staged kernel at v4's shapes: 5.8 VALU instructions per MFMA (the per-row pixel bookkeeping and
// 64-bit addresses of load_tile), a third of the LDS cycles lost to bank conflicts, MFMA pipe 17 % busy.  Here
//   * tiles are [64 pixel rows][BC channels] with UNPADDED rows, filled by buffer_load ... lds in 1 KB pieces
//     (1024 / row-bytes rows each); rows beyond the split, taps outside the image and channels beyond the matrix are
//     out-of-range offsets = zero fill;
//   * bank conflicts of ds_read_b64_tr_b16 are avoided by an XOR of the 32-byte column-pair index with a key of the
//     row: a half-wave reads rows {r..r+3, r+8..r+11}, 32 bytes of one column pair each, and 256 bytes of distinct
//     banks need, per row size,  256 B: key = (row&3) | ((row>>3)&1)<<2;  128 B: ((row>>1)&1) | ((row>>3)&1)<<1;
//     64 B: (row>>3)&1.  For the rows a lane reads the key is a per-lane constant, so every LDS address of the MFMA
//     phase is one per-lane constant plus an immediate;
//   * the dY pieces advance linearly with the pixel (one add per piece and step); only the tap-shifted X rows keep
//     the (n, oh, ow) bookkeeping, once per piece instead of once per 16-byte load.
template <int RB>
__device__ __forceinline__ int wg_key(int row) {
    if constexpr (RB >= 256) return (row & 3) | (((row >> 3) & 1) << 2);
    else if constexpr (RB == 128) return ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
    else return (row >> 3) & 1;
}

// LD = 1: four extra waves (one per SIMD) issue every DMA piece and carry the pixel bookkeeping of the tap-shifted X rows; the four
// compute waves only read fragments and issue MFMAs.  (In the 4-wave form a wave spends ~100 cycles issuing each of its 8 pieces
// per 64-pixel step against 512 cycles of MFMAs: with two blocks per CU the MFMA pipe cannot exceed ~40 % -- rocprofv3 counted
// 27-33 %.)  Two blocks per CU in both forms (64 KB of LDS each; LD = 1 needs <= 128 VGPRs).
// NST > 2 (with LD = 1): a ring of NST stages of BKP_ pixels with COUNTED vmcnt -- the loaders keep two tiles in flight across
// every barrier instead of waiting for the tile they just issued (the protocol of conv_pipe.hip's loader variant); 4 stages of 32
// pixels take the same 64 KB as 2 stages of 64, so two blocks per CU remain.
// NCW = 8 (with LD = 1, NST = 3): eight compute waves, 4 over channels x 2 over columns, each on the same 64 x 64 block as in
// the 4-wave forms -- a 256 x 128 block tile, ONE block per CU (144 KB of LDS).  The 128 x 128 tile needs 64 B/clk of L2 -> LDS
// traffic per CU at full MFMA rate, which is all a CU's load path delivers; 256 x 128 needs 48.
template <typename T, int BCO, int BCI, int LD = 0, int BKP_ = 64, int NST = 2, int NCW = 4>
__global__ void __launch_bounds__(64 * NCW + 256 * LD, NCW == 8 ? 3 : (LD ? 4 : 2)) wgrad_dma_kernel(const WgradParams p) {
    constexpr int BKP = BKP_;
    constexpr int WCO = (NCW == 8 && BCO == 256) ? 4 : 2;    // compute waves over the output channels ...
    constexpr int WCI = NCW / WCO;                           // ... and over the (tap, ci) columns
    static_assert(NST == 2 || (LD == 1 && (NST == 3 || NST == 4)), "ring variants");
    static_assert(NCW == 4 || (NCW == 8 && LD == 1 && NST > 2 && ((BCO == 256 && BCI == 128) || (BCO == 128 && BCI == 256))), "8-wave variants");
    constexpr int RBA = BCO * 2, RBB = BCI * 2;          // row bytes
    constexpr int CPA = RBA / 16, CPB = RBB / 16;        // 16-byte chunks per row
    constexpr int RPA = 64 / CPA, RPB = 64 / CPB;        // rows per 1 KB piece
    constexpr int NPA = BKP / RPA / 4, NPB = BKP / RPB / 4;  // pieces per wave and K step
    constexpr int STAGE = BKP * (RBA + RBB);
    constexpr int TI = BCO / (16 * WCO), TJ = BCI / (16 * WCI);
    static_assert(NPA >= 1 && NPB >= 1, "tile too narrow");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_loader = LD && wave_all >= NCW;        // wave-uniform role
    const bool stages = !LD || is_loader;
    const int wave = is_loader ? wave_all - NCW : wave_all;   // index within the role (loaders: 0..3)
    const int wi = wave % WCO, wj = wave / WCO;
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, slot = bid >> 3;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int ct = lid % p.ncol_tiles, rest = lid / p.ncol_tiles;
    const int col0 = ct * BCI, co0 = (rest % p.nco_tiles) * BCO, sp = rest / p.nco_tiles;
    const int pix_begin = sp * p.pps;
    const int pix_end = min(p.M, pix_begin + p.pps);
    const int wave_u = wave;

    // ---- MFMA-phase LDS addresses: per-lane constants (the swizzle key of the rows a lane reads does not depend on kk) ----
    const int q16 = lane & 15, g = lane >> 4;
    const int prow0 = g * 8 + (q16 >> 2);
    unsigned a_col[TI], b_col[TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int cbyte = (wi * (BCO / WCO) + i * 16 + ((q16 & 3) << 2)) * 2;
        a_col[i] = (unsigned)(prow0 * RBA + ((((cbyte >> 4) ^ (wg_key<RBA>(prow0) << 1)) << 4) | (cbyte & 15)));
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int cbyte = (wj * (BCI / WCI) + j * 16 + ((q16 & 3) << 2)) * 2;
        b_col[j] = (unsigned)(BKP * RBA + prow0 * RBB + ((((cbyte >> 4) ^ (wg_key<RBB>(prow0) << 1)) << 4) | (cbyte & 15)));
    }

    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkt = pix_end > pix_begin ? (pix_end - pix_begin + BKP - 1) / BKP : 0;
    auto mma_tile = [&](int cur) {
        const unsigned char* st = smem + cur * STAGE;
#pragma unroll
        for (int kk = 0; kk < BKP / 32; ++kk) {
            typename WMma<T>::frag a[TI], b[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const unsigned char* ptr = st + a_col[i] + kk * 32 * RBA;
                a[i] = tr16_pair<T>(ptr, ptr + 4 * RBA);
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const unsigned char* ptr = st + b_col[j] + kk * 32 * RBB;
                b[j] = tr16_pair<T>(ptr, ptr + 4 * RBB);
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = WMma<T>::mma(a[i], b[j], acc[i][j]);
        }
    };
    if (stages) {
        // (everything the DMA issue needs lives in this scope: in the LD form the compute waves never hold it in registers)
        // descriptors: dY is cut at the end of this block's pixel range (rows beyond it read as zero)
        const auto rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (unsigned)pix_end * (unsigned)p.lddy * 2u, 0x00020000);
        const auto rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);

        // ---- A pieces (dY): piece k of this wave = tile rows (wave + 4k) * RPA + lane / CPA ----------------------------
        unsigned a_off[NPA];
#pragma unroll
        for (int k = 0; k < NPA; ++k) {
            const int row = (wave + 4 * k) * RPA + lane / CPA;
            const int lchunk = (lane % CPA) ^ (wg_key<RBA>(row) << 1);
            const int co = co0 + lchunk * 8;
            a_off[k] = co < p.Co ? ((unsigned)(pix_begin + row) * (unsigned)p.lddy + (unsigned)co) * 2u : 0xFFFFFFFFu;
        }
        const unsigned a_step = (unsigned)BKP * (unsigned)p.lddy * 2u;
        // ---- B pieces (X shifted by the tap of the lane's column): pixel bookkeeping per piece -----------------------------
        int bn[NPB], boh[NPB], bow[NPB], b_dh[NPB], b_dw[NPB];
        unsigned b_ci[NPB];
        bool b_cok[NPB];
        const int ohw = p.OH * p.OW;
#pragma unroll
        for (int k = 0; k < NPB; ++k) {
            const int row = (wave + 4 * k) * RPB + lane / CPB;
            const int lchunk = (lane % CPB) ^ (wg_key<RBB>(row) << 1);
            const int col = col0 + lchunk * 8;
            b_cok[k] = col < p.Ncols;
            const int tap = b_cok[k] ? col / p.Ci : 0;
            b_ci[k] = (unsigned)(col - tap * p.Ci) * 2u;
            const int kh = tap / p.ks;
            b_dh[k] = kh - p.pad;
            b_dw[k] = tap - kh * p.ks - p.pad;
            const int m = pix_begin + row;
            const int n = m / ohw, rem = m - n * ohw;
            bn[k] = n; boh[k] = rem / p.OW; bow[k] = rem - boh[k] * p.OW;
        }
        const unsigned pixb = (unsigned)p.ldx * 2u;
        const int adv_h = BKP / p.OW, adv_w = BKP - adv_h * p.OW;

        auto load_tile = [&](int kt, int stage) {
            unsigned char* as_w = smem + stage * STAGE + wave_u * 1024;
            unsigned char* bs_w = smem + stage * STAGE + BKP * RBA + wave_u * 1024;
#pragma unroll
            for (int k = 0; k < NPA; ++k) {
                const unsigned v = a_off[k];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(as_w + k * 4096), 16, v, 0, 0, 0);
                a_off[k] = v == 0xFFFFFFFFu ? v : v + a_step;
            }
#pragma unroll
            for (int k = 0; k < NPB; ++k) {
                const int m = pix_begin + kt * BKP + (wave + 4 * k) * RPB + lane / CPB;
                const int xh = boh[k] * p.stride + b_dh[k], xw = bow[k] * p.stride + b_dw[k];
                const bool ok = b_cok[k] & (m < pix_end) & ((unsigned)xh < (unsigned)p.XH) & ((unsigned)xw < (unsigned)p.XW);
                const unsigned v = ok ? (unsigned)((bn[k] * p.XH + xh) * p.XW + xw) * pixb + b_ci[k] : 0xFFFFFFFFu;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void*)(bs_w + k * 4096), 16, v, 0, 0, 0);
                // advance the row's pixel by BKP = adv_h * OW + adv_w without loops (the host guarantees adv_h + 1 <= 2 * OH)
                bow[k] += adv_w;
                const bool cw = bow[k] >= p.OW;
                bow[k] -= cw ? p.OW : 0;
                boh[k] += adv_h + (cw ? 1 : 0);
                const bool c1 = boh[k] >= p.OH;
                boh[k] -= c1 ? p.OH : 0;
                const bool c2 = boh[k] >= p.OH;
                boh[k] -= c2 ? p.OH : 0;
                bn[k] += (c1 ? 1 : 0) + (c2 ? 1 : 0);
            }
        };


        if constexpr (NST > 2) {
            // tile kt lives in stage kt % NST.  At the barrier that opens step kt the loaders have retired tile kt (in-order
            // return: vmcnt <= the pieces of the NST - 2 newer tiles still flying) and the compute waves are done with tile
            // kt - 1, whose stage takes tile kt + NST - 1 while they multiply tile kt.
            constexpr int NPL = NPA + NPB;
#pragma unroll
            for (int t = 0; t < NST - 1; ++t)
                if (t < nkt) load_tile(t, t);
            int fill = NST - 1;                 // stage of the next tile to issue
            for (int kt = 0; kt < nkt; ++kt) {
                if (NST == 4 && kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPL) : "memory");
                else if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPL) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (kt + NST - 1 < nkt) load_tile(kt + NST - 1, fill);
                fill = fill == NST - 1 ? 0 : fill + 1;
            }
            __builtin_amdgcn_s_barrier();     // closes the ring: matches the compute waves' barrier before their epilogue
            return;
        } else {
            if (nkt > 0) load_tile(0, 0);
            __syncthreads();
            for (int kt = 0; kt < nkt; ++kt) {
                const int cur = kt & 1;
                if (kt + 1 < nkt) load_tile(kt + 1, cur ^ 1);
                if constexpr (!LD) mma_tile(cur);
                __syncthreads();
            }
            if constexpr (LD) return;     // (the epilogue below has no block-wide barrier)
        }
    } else {
        if constexpr (NST > 2) {
            int cs = 0;
            for (int kt = 0; kt < nkt; ++kt) {
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                mma_tile(cs);
                cs = cs == NST - 1 ? 0 : cs + 1;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();     // every wave is done reading the ring: the epilogue reuses it
            asm volatile("" ::: "memory");
        } else {
            __syncthreads();
            for (int kt = 0; kt < nkt; ++kt) {
                mma_tile(kt & 1);
                __syncthreads();
            }
        }
    }

    if constexpr (NCW == 8 || (BCO == 128 && BCI == 128)) {
        // The slab tile leaves through LDS (the ring is free now): each wave transposes its 64 x 64 accumulator block into
        // [row][col] fp32 rows (stride 68 words: the four row groups of a fragment land on disjoint banks) and stores 16 bytes
        // per lane, 256 contiguous bytes per row -- 16 store instructions per lane instead of 64 four-byte ones in 64-byte
        // runs.  (Atomic mode adds word by word as before.)
        if (!p.atomic) {
            constexpr int RS = 68;
            float* wt = reinterpret_cast<float*>(smem) + wave * (32 * RS);
            static_assert(NCW * 32 * RS * 4 <= NST * STAGE, "the waves' tiles (32 rows at a time) fit in the ring");
            float* slab = p.part + (size_t)sp * p.CoRows * p.Ncols;
            const int c4 = (lane & 15) * 4, r0 = lane >> 4;
            const int col = col0 + wj * 64 + c4;
#pragma unroll
            for (int h = 0; h < TI / 2; ++h) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) wt[(i * 16 + g * 4 + r) * RS + j * 16 + q16] = acc[2 * h + i][j][r];
                // (wave-private tile: the wave's ds_writes are ordered before its ds_reads by lgkmcnt, no barrier)
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int row = it * 4 + r0;
                    const int co = co0 + wi * 64 + h * 32 + row;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(wt + row * RS + c4);
                    if (co >= p.CoRows || col >= p.Ncols) continue;
                    float* dst = slab + (size_t)co * p.Ncols + col;
                    if (col + 3 < p.Ncols && (p.Ncols & 3) == 0) *reinterpret_cast<f32x4*>(dst) = v;
                    else
                        for (int e = 0; e < 4 && col + e < p.Ncols; ++e) dst[e] = v[e];
                }
            }
            return;
        }
    }
    if constexpr (NCW == 4) wgrad_store<TI, TJ, BCO, BCI>(p, acc, sp, co0, col0, wi, wj, q16, g);     // (8 waves: slab stores only)
}

template <typename T, int BCO, int BCI, int LD = 0, int BKP = 64, int NST = 2, int NCW = 4>
int launch_dma(const WgradParams& p, int split, hipStream_t s) {
    constexpr int smem = NST * BKP * (BCO * 2 + BCI * 2);
    static unsigned long long attr_done = 0;      // bit d: set for HIP device d
    if (cy_first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_dma_kernel<T, BCO, BCI, LD, BKP, NST, NCW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
    WgradParams q = p;
    q.ncol_tiles = (p.Ncols + BCI - 1) / BCI;
    q.nco_tiles = (p.CoRows + BCO - 1) / BCO;
    hipLaunchKernelGGL((wgrad_dma_kernel<T, BCO, BCI, LD, BKP, NST, NCW>), dim3(q.ncol_tiles * q.nco_tiles * split), dim3(64 * NCW + 256 * LD), smem, s, q);
    CY_LAUNCH_CHECK();
    return 0;
}

template <typename T, int BCO, int BCI, bool USE_TR>
int launch(const WgradParams& p, int split, hipStream_t s) {
    constexpr int BKP = WTraits<T>::BKP;
    constexpr int smem = 2 * BKP * (BCO * (int)sizeof(T) + 32 + BCI * (int)sizeof(T) + 32);
    static unsigned long long attr_done = 0;      // bit d: set for HIP device d
    if (cy_first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<T, BCO, BCI, USE_TR>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
    WgradParams q = p;
    q.ncol_tiles = (p.Ncols + BCI - 1) / BCI;
    q.nco_tiles = (p.CoRows + BCO - 1) / BCO;
    hipLaunchKernelGGL((wgrad_kernel<T, BCO, BCI, USE_TR>), dim3(q.ncol_tiles * q.nco_tiles * split), dim3(256), smem, s, q);
    CY_LAUNCH_CHECK();
    return 0;
}

inline int tile_of(int c) { return c > 64 ? 128 : (c > 32 ? 64 : 32); }

template <typename T>
int dispatch_dma(const WgradParams& p, int split, int cap, hipStream_t s) {
    // cap = 64: 64 x 64 tiles where 128 x 128 would be chosen -- four times the tiles, so a quarter of the split-K factor (and of
    // the slab traffic) for the same number of blocks; the engine times both (layers with a small weight matrix and many pixels)
    int bco = tile_of(p.CoRows), bci = tile_of(p.Ncols);
    if (cap == 64) { if (bco > 64) bco = 64; if (bci > 64) bci = 64; }
    // CY_WGRAD_LOADERS: 0 = four self-staging waves everywhere, 1 = loader waves for the 128 x 128 tile only, 2 = for every tile with
    // at least 64 channels on both sides, 4 (default) = 2 + the 256 x 128 tile on eight compute waves for layers with >= 256 output
    // channels, 5 = 4 + the 128 x 256 tile for the 128-channel layers, 3 = 2 with the 128 x 128 tile on a four-stage ring.
    // Measured, each level with its own tuned split table, same-box pairs: 0 -> 1 -> 2: 821 -> 830 -> 832 images/s; 2 -> 4: 850.5 ->
    // 863.8; 3 is 0.3 % behind 2; 5 is 0.85 % behind 4 although its kernels alone are faster (128 -> 128 3x3 @76x76: 47 -> 41 us): the
    // weight gradients run beside the trunk's dgrad / BN kernels, and one-block-per-CU tiles at the 76 x 76 stage get in their way.
    static int loaders = -1;
    if (loaders < 0) { const char* e = getenv("CY_WGRAD_LOADERS"); loaders = e ? atoi(e) : 4; }
    if (loaders >= 4 && cap != 64 && !p.atomic) {
        if (p.CoRows >= 256 && bci == 128) return launch_dma<T, 256, 128, 1, 64, 3, 8>(p, split, s);
        if (loaders == 5 && bco == 128 && p.Ncols >= 256) return launch_dma<T, 128, 256, 1, 64, 3, 8>(p, split, s);
    }
    if (loaders == 3 && bco == 128 && bci == 128) return launch_dma<T, 128, 128, 1, 32, 4>(p, split, s);     // experiment: counted-vmcnt ring
    if (loaders && bco == 128 && bci == 128) return launch_dma<T, 128, 128, 1>(p, split, s);
    if (loaders >= 2) {
        if (bco == 128 && bci == 64) return launch_dma<T, 128, 64, 1>(p, split, s);
        if (bco == 64 && bci == 128) return launch_dma<T, 64, 128, 1>(p, split, s);
        if (bco == 64 && bci == 64) return launch_dma<T, 64, 64, 1>(p, split, s);
    }
#define CY_WD(A, B) \
    if (bco == A && bci == B) return launch_dma<T, A, B>(p, split, s);
    CY_WD(128, 128) CY_WD(128, 64) CY_WD(128, 32) CY_WD(64, 128) CY_WD(64, 64) CY_WD(64, 32) CY_WD(32, 128) CY_WD(32, 64)
    CY_WD(32, 32)
#undef CY_WD
    return CY_ERR_ARG;
}

template <typename T, bool USE_TR>
int dispatch(const WgradParams& p, int split, hipStream_t s) {
    const int bco = tile_of(p.CoRows), bci = tile_of(p.Ncols);
#define CY_W(A, B) \
    if (bco == A && bci == B) return launch<T, A, B, USE_TR>(p, split, s);
    CY_W(128, 128) CY_W(128, 64) CY_W(128, 32) CY_W(64, 128) CY_W(64, 64) CY_W(64, 32) CY_W(32, 128) CY_W(32, 64)
    CY_W(32, 32)
#undef CY_W
    return CY_ERR_ARG;
}

// Fold the split-K slabs: block = 64 consecutive output elements x 4 waves striding over the slabs (every wave
// reads 256 contiguous bytes of one slab per step), combined through LDS; then scatter into the OIHW gradient.
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ part, int split, int CoRows, int CiPad,
                                                          int ks, int Co, int Ci, float scale, int accumulate,
                                                          float* __restrict__ grad) {
    __shared__ float red[4][64];
    const int kk = ks * ks;
    const long total = (long)Co * kk * Ci;   // walked in slab order (co, tap, ci)
    const int ncols = kk * CiPad;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long idx = (long)blockIdx.x * 64 + lane;
    float s = 0.f;
    size_t src = 0, dst = 0;
    if (idx < total) {
        const int ci = (int)(idx % Ci);
        const long t = idx / Ci;
        const int tap = (int)(t % kk), co = (int)(t / kk);
        src = (size_t)co * ncols + tap * CiPad + ci;
        dst = ((size_t)co * Ci + ci) * kk + tap;
        const size_t slab = (size_t)CoRows * ncols;
        for (int sp = w; sp < split; sp += 4) s += part[(size_t)sp * slab + src];
    }
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && idx < total) {
        const float v = scale * (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
        grad[dst] = v + (accumulate ? grad[dst] : 0.f);
    }
}

// table-driven fold: block -> (descriptor, first (co, ci) pair / 256); one thread per (co, ci) pair walks all slabs of its
// ks*ks taps (consecutive threads = consecutive ci: 256-byte reads per wave, tap and slab), then the block transposes its
// [256 pairs][kk] sums through LDS so that the OIHW gradient -- 4 bytes every kk floats from a pair's point of view -- is
// read and written as ONE contiguous run of 256*kk floats.  The earlier thread-per-output version scattered 4-byte
// writes at a 36-byte stride: rocprofv3 counted 1.75 GB of HBM traffic per launch against 0.87 GB of slabs (DESIGN.md
// section 9, PMC table).
template <int KK>
__device__ __forceinline__ void fold_pairs(const cy_reduce_desc& d, long first_pair, float scale, int accumulate,
                                           float* lds) {
    // d.lanes threads share a pair and split the slabs between them (small layers with a large split-K factor have
    // few pairs and a long serial chain per thread otherwise: 128 slabs x 9 taps was 0.3 ms for ONE block, exposed at
    // the end of backward); the partial sums meet in LDS in a fixed order
    const int lanes = d.lanes, PB = 256 / lanes;
    const int pl = threadIdx.x % PB, sl = threadIdx.x / PB;
    const long npairs = (long)d.Co * d.Ci;
    const int ncols = KK * d.CiPad;
    const size_t slab = (size_t)d.CoRows * ncols;
    const long p = first_pair + pl;
    float acc[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) acc[t] = 0.f;
    if (p < npairs) {
        const int ci = (int)(p % d.Ci), co = (int)(p / d.Ci);
        const float* src = d.part + (size_t)co * ncols + ci;
        // (four slabs' loads in flight per trip: the sums stay in slab order, acc[t] is one chain either way)
#pragma unroll 4
        for (int sp = sl; sp < d.split; sp += lanes) {
#pragma unroll
            for (int t = 0; t < KK; ++t) acc[t] += src[(size_t)sp * slab + t * d.CiPad];
        }
        if (d.flags & 1) {       // atomically accumulated slab (cy_conv_wgrad, atomic mode): leave it zeroed for the next step
            float* z = const_cast<float*>(src);
            for (int sp = sl; sp < d.split; sp += lanes) {
#pragma unroll
                for (int t = 0; t < KK; ++t) z[(size_t)sp * slab + t * d.CiPad] = 0.f;
            }
        }
    }
    if (KK == 1 && lanes == 1) {
        if (p < npairs) d.grad[p] = scale * acc[0] + (accumulate ? d.grad[p] : 0.f);
        return;
    }
#pragma unroll
    for (int t = 0; t < KK; ++t) lds[threadIdx.x * KK + t] = acc[t];     // [sl][pl][t]
    __syncthreads();
    const long total = npairs * KK, o0 = first_pair * KK;
    const int span = PB * KK;                       // this block's contiguous run of the OIHW gradient
    for (int o = threadIdx.x; o < span; o += 256) {
        if (o0 + o >= total) break;
        float v = lds[o];
        for (int l = 1; l < lanes; ++l) v += lds[l * span + o];
        d.grad[o0 + o] = scale * v + (accumulate ? d.grad[o0 + o] : 0.f);
    }
}

__global__ void __launch_bounds__(256) wgrad_reduce_multi_kernel(const cy_reduce_desc* __restrict__ desc,
                                                                const int* __restrict__ blocks, float scale,
                                                                int accumulate) {
    __shared__ float lds[256 * 9];
    const cy_reduce_desc d = desc[blocks[2 * blockIdx.x]];
    const long first = (long)blocks[2 * blockIdx.x + 1] * 32;
    if (d.ks == 1) fold_pairs<1>(d, first, scale, accumulate, lds);
    else if (d.ks == 3) fold_pairs<9>(d, first, scale, accumulate, lds);
    else {   // other kernel sizes: one pair per thread, strided writes
        const int kk = d.ks * d.ks;
        const long p = first + threadIdx.x;
        if (p >= (long)d.Co * d.Ci) return;
        const int ci = (int)(p % d.Ci), co = (int)(p / d.Ci);
        const int ncols = kk * d.CiPad;
        const size_t slab = (size_t)d.CoRows * ncols;
        for (int t = 0; t < kk; ++t) {
            float sum = 0.f;
            for (int sp = 0; sp < d.split; ++sp) {
                float* q = const_cast<float*>(d.part) + (size_t)sp * slab + (size_t)co * ncols + t * d.CiPad + ci;
                sum += *q;
                if (d.flags & 1) *q = 0.f;
            }
            d.grad[p * kk + t] = scale * sum + (accumulate ? d.grad[p * kk + t] : 0.f);
        }
    }
}

}  // namespace

extern "C" int cy_conv_wgrad_split(int M, int Co, int Ci, int ks) {
    CY_ENTER();
    const int ncols = ks * ks * Ci;
    const long tiles = (long)((Co + tile_of(Co) - 1) / tile_of(Co)) * ((ncols + tile_of(ncols) - 1) / tile_of(ncols));
    const long target = 768, minpix = 512;   // ~3 blocks per CU; at least minpix / 64 K steps per block (the engine then times the neighbours)
    long split = (target + tiles - 1) / tiles;
    const long max_by_work = (M + minpix - 1) / minpix;  // at least minpix/64 K steps per block
    if (split > max_by_work) split = max_by_work;
    const long slab = (long)Co * ncols * 4;
    while (split > 1 && split * slab > (256L << 20)) --split;
    return (int)(split < 1 ? 1 : split);
}

extern "C" int cy_conv_wgrad(const void* dy, int N, int OH, int OW, int Co, int lddy, const void* x, int XH, int XW,
                             int Ci, int ldx, int ks, int stride, int pad, int dtype, float* part, int split,
                             int use_tr, cy_stream_t s) {
    CY_ENTER();
    const int ch = dtype == CY_F32 ? 4 : 8;
    if (!dy || !x || !part || split < 1 || (dtype != CY_F16 && dtype != CY_BF16 && dtype != CY_F32)) return CY_ERR_ARG;
    if (Co % ch || Ci % ch || lddy % ch || ldx % ch || ks < 1 || ks > 3) return CY_ERR_ARG;
    WgradParams p;
    p.dy = (const unsigned char*)dy; p.x = (const unsigned char*)x; p.part = part;
    p.N = N; p.OH = OH; p.OW = OW; p.Co = Co; p.lddy = lddy;
    p.XH = XH; p.XW = XW; p.Ci = Ci; p.ldx = ldx; p.ks = ks; p.stride = stride; p.pad = pad;
    p.M = N * OH * OW; p.Ncols = ks * ks * Ci; p.CoRows = Co;
    const int bkp = dtype == CY_F32 ? 32 : 64;
    p.pps = (((p.M + split - 1) / split) + bkp - 1) / bkp * bkp;
    p.x_bytes = 0;
    p.atomic = (use_tr & 4) ? 1 : 0;
    const int cap = (use_tr & 8) ? 64 : 128;
    use_tr &= 3;
    if (dtype == CY_F32) return dispatch<float, false>(p, split, cy_s(s));
    {   // direct-to-LDS kernel; use_tr = 2 forces the register-staged kernel (A/B runs), as do offsets beyond 32 bits
        const size_t xb = (((size_t)N * XH * XW - 1) * ldx + Ci) * 2, ab = ((size_t)p.M + 128) * lddy * 2;
        if (use_tr == 1 && xb < 0xFFFFFF00ull && ab < 0xFFFFFF00ull && 64 / OW + 1 <= 2 * OH) {
            p.x_bytes = (unsigned)xb;
            return dtype == CY_F16 ? dispatch_dma<f16>(p, split, cap, cy_s(s)) : dispatch_dma<bf16>(p, split, cap, cy_s(s));
        }
    }
    if (dtype == CY_BF16) return use_tr ? dispatch<bf16, true>(p, split, cy_s(s)) : dispatch<bf16, false>(p, split, cy_s(s));
    return use_tr ? dispatch<f16, true>(p, split, cy_s(s)) : dispatch<f16, false>(p, split, cy_s(s));
}

extern "C" int cy_conv_wgrad_bn(const void* g, int N, int OH, int OW, int Co, int ldg, const void* raw, int ldraw, const void* x,
                                int XH, int XW, int Ci, int ldx, int ks, int stride, int pad, int dtype, const float* mean,
                                const float* invstd, const float* scale, const float* shift, const float* part_bins, int rows,
                                float* ggamma, float* gbeta, float gscale, float* zero_table, int zero_n, int act, float* part,
                                int split, cy_stream_t s) {
    CY_ENTER();
    if (!g || !raw || !x || !part || !mean || !invstd || !scale || !shift || !part_bins || split < 1 || rows != CY_STAT_BINS ||
        (zero_n > 0 && !zero_table) || zero_table == part_bins)
        return CY_ERR_ARG;
    if (dtype != CY_F16 && dtype != CY_BF16) return CY_ERR_UNSUPPORTED;
    if (Co % 8 || Co > 32 || Ci % 8 || ldg % 8 || ldraw % 8 || ldx % 8 || ks < 1 || ks > 3) return CY_ERR_UNSUPPORTED;
    WgradParams p;
    p.dy = (const unsigned char*)g; p.x = (const unsigned char*)x; p.part = part;
    p.N = N; p.OH = OH; p.OW = OW; p.Co = Co; p.lddy = ldg;
    p.XH = XH; p.XW = XW; p.Ci = Ci; p.ldx = ldx; p.ks = ks; p.stride = stride; p.pad = pad;
    p.M = N * OH * OW; p.Ncols = ks * ks * Ci; p.CoRows = Co;
    p.pps = (((p.M + split - 1) / split) + 63) / 64 * 64;
    p.x_bytes = 0; p.atomic = 0;
    p.ncol_tiles = (p.Ncols + 127) / 128; p.nco_tiles = 1;
    WgradBnParams f;
    f.raw = (const unsigned char*)raw; f.ldraw = ldraw; f.mean = mean; f.invstd = invstd; f.scale = scale; f.shift = shift;
    f.bins = part_bins; f.ggamma = ggamma; f.gbeta = gbeta; f.gscale = gscale; f.zero_table = zero_table; f.zero_n = zero_n;
    constexpr int smem = 2 * 64 * ((32 * 2 + 32) + (128 * 2 + 32));
    const dim3 grid((unsigned)(p.ncol_tiles * split));
#define CY_WBN(T, A) hipLaunchKernelGGL((wgrad_bn_kernel<T, A>), grid, dim3(256), smem, cy_s(s), p, f);
#define CY_WBN_ACT(T)                                        \
    if (act == CY_ACT_MISH) { CY_WBN(T, CY_ACT_MISH) }       \
    else if (act == CY_ACT_LEAKY) { CY_WBN(T, CY_ACT_LEAKY) } \
    else { CY_WBN(T, CY_ACT_LINEAR) }
    if (dtype == CY_F16) { CY_WBN_ACT(f16) } else { CY_WBN_ACT(bf16) }
#undef CY_WBN
#undef CY_WBN_ACT
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_wgrad_reduce(const float* part, int split, int CoRows, int CiPad, int ks, int Co, int Ci, float scale,
                               int accumulate, float* grad, cy_stream_t s) {
    CY_ENTER();
    if (!part || !grad || split < 1 || Co > CoRows || Ci > CiPad) return CY_ERR_ARG;
    const long total = (long)Co * Ci * ks * ks;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, cy_s(s), part, split, CoRows,
                       CiPad, ks, Co, Ci, scale, accumulate, grad);
    CY_LAUNCH_CHECK();
    return 0;
}

extern "C" int cy_wgrad_reduce_multi(const cy_reduce_desc* desc, const int32_t* blocks, int nblocks, float scale,
                                     int accumulate, cy_stream_t s) {
    CY_ENTER();
    if (!desc || !blocks || nblocks < 1) return CY_ERR_ARG;   // (desc[i].lanes in {1, 2, 4, 8}: the table builder's contract)
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(nblocks), dim3(256), 0, cy_s(s), desc, blocks, scale, accumulate);
    CY_LAUNCH_CHECK();
    return 0;
}
