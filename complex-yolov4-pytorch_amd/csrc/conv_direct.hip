// Direct 3x3 convolution for the small-Cin forward layers of the Darknet stack (reference darknet2pytorch.py:247-278):
// 3 -> 32 @608 (Cin padded to 8), 32 -> 64 stride 2 @608 -> 304 and 32 -> 64 @304.  K = 9 * Cin is 72 or 288: per pixel
// the GEMM is a handful of MFMAs and the layers are pure streaming -- the implicit-GEMM kernels spend their time on per-block
// prologues, K-step barriers and a 9-fold gather through the load path (rocprofv3, batch 32 inference: 752 + 2 x 636 us for
// these three launches, 18 % of the step, against HBM floors of 190 / 230 / 115 us).
//
// One persistent block per CU slot loops over output tiles of TH x TW pixels of one image:
//   * the packed weights [Cout][9 * Cin] sit in LDS for the life of the block (rows padded by 16 bytes: conflict-free A reads);
//   * the input patch ((TH-1) S + 3) x ((TW-1) S + 3) x Cin is read ONCE, 16 bytes per lane, zero outside the image;
//   * a wave computes 32-pixel row segments: the B fragment of MFMA step j is one ds_read_b128 of the patch at the pixel
//     shifted by the step's tap (k = tap * Cin + c: a 16-byte chunk = 8 channels of one tap) -- no im2col image;
//   * epilogue like conv_pipe.hip: BN batch statistics (32-lane sums, one atomic per channel, moment and block), or the
//     eval-mode affine + activation (+ shortcut); the accumulators are transposed through LDS and stored 16 bytes per lane.
#include <string.h>

#include "igemm_common.hpp"

namespace {
using namespace cyk;

template <typename T, int CIN, int COUT, int S, int TH, int TW, int SPW>
struct DirectCfg {
    static constexpr int NSEG = TH * (TW / 32), NWAVE = NSEG / SPW, NT = NWAVE * 64;
    static constexpr int PR = (TH - 1) * S + 3, PC = (TW - 1) * S + 3;   // patch rows / columns
    static constexpr int PXB = CIN * 2, CPP = CIN / 8;                     // bytes / 16-byte chunks per patch pixel
    static constexpr int K = 9 * CIN, KS = (K + 15) / 16;                  // reduction length, MFMA steps
    static constexpr int WROW = K * 2 + 16;                                // weight row in LDS (one chunk of padding, zeroed)
    static constexpr int NCB = COUT / 32;
    static constexpr int W_BYTES = COUT * WROW;
    static constexpr int SROW = COUT * 4 + 16;                             // fp32 store-tile row (padded: spreads the banks)
    static constexpr int ST_BYTES = NWAVE * 32 * SROW;                     // wave-private [32 pixels][COUT] fp32 tiles ...
    static constexpr int PATCH_BYTES = (PR * PC * PXB + 15) / 16 * 16;
    static constexpr int P_BYTES = PATCH_BYTES > ST_BYTES ? PATCH_BYTES : ST_BYTES;   // ... which reuse the patch's LDS
    static constexpr int RED_BYTES = NWAVE * 2 * COUT * 4 + 2 * COUT * 4;   // block reduction of the statistics + (scale, shift)
    static constexpr int SMEM = W_BYTES + P_BYTES + RED_BYTES;
    static constexpr int BLOCKS_PER_CU = SMEM > 40 * 1024 ? 2 : 4;
    static_assert(NSEG % SPW == 0 && TW % 32 == 0 && (CIN == 8 || CIN == 32) && COUT % 32 == 0, "tile layout");
};

// physical 16-byte chunk of logical chunk c of patch column pcol (Cin = 32: four chunks per pixel, spread so that 16
// consecutive columns hit 16 distinct bank groups)
template <int CPP>
__device__ __forceinline__ int pswz(int c, int pcol) {
    return CPP == 1 ? 0 : (c ^ (pcol & 3) ^ ((pcol >> 2) & 3));
}

template <typename T, int CIN, int COUT, int S, int TH, int TW, int SPW>
__global__ void __launch_bounds__((TH * (TW / 32) / SPW) * 64, (DirectCfg<T, CIN, COUT, S, TH, TW, SPW>::BLOCKS_PER_CU))
    direct3x3_kernel(const IgemmParams p) {
    typedef DirectCfg<T, CIN, COUT, S, TH, TW, SPW> C;
    typedef typename Mma32<T>::frag frag;
    typedef T tx4 __attribute__((ext_vector_type(4)));
    typedef T tx8 __attribute__((ext_vector_type(8)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wl = smem;
    unsigned char* patch = smem + C::W_BYTES;
    unsigned char* stage = patch;          // reused once every wave is done reading the patch
    float* red = reinterpret_cast<float*>(patch + C::P_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;

    // ---- weights -> LDS (once per block) ---------------------------------------------------------------------------------
    {
        constexpr int CPRW = C::K * 2 / 16;          // 16-byte chunks per weight row
        for (int idx = tid; idx < COUT * (CPRW + 1); idx += C::NT) {
            const int row = idx / (CPRW + 1), ch = idx - row * (CPRW + 1);
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (ch < CPRW && row < p.wrows) v = *reinterpret_cast<const u32x4*>(p.w + (size_t)row * (C::K * 2) + ch * 16);
            *reinterpret_cast<u32x4*>(wl + row * C::WROW + ch * 16) = v;
        }
    }
    const int tiles_x = (p.OW + TW - 1) / TW, tiles_y = (p.OH + TH - 1) / TH;
    const int ntiles = p.N * tiles_y * tiles_x;
    const bool stats = (p.flags & CY_CONV_STATS) != 0, affine = (p.flags & CY_CONV_AFFINE_ACT) != 0;
    constexpr int NSB = (COUT + 63) / 64;
    float ssum[NSB], qsum[NSB];            // lane c: channel sb*64 + c
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) { ssum[sb] = 0.f; qsum[sb] = 0.f; }
    // eval epilogue: the BN affine of the layer, in LDS behind the reduction scratch (read per accumulator register)
    float* asc = red + C::NWAVE * 2 * COUT;
    float* ash = asc + COUT;
    if (affine)
        for (int c = tid; c < COUT; c += C::NT) { asc[c] = p.aff_scale[min(c, p.OC - 1)]; ash[c] = p.aff_shift[min(c, p.OC - 1)]; }
    const size_t pixg = (size_t)p.ldg * 2, pixo = (size_t)p.ldo * 2;

    // The patch of the NEXT tile is requested into registers before this tile's MFMAs and epilogue and written to LDS after
    // them: its memory latency hides behind a whole tile of work (two blocks per CU alone could not hide it).
    constexpr int NCH = C::PR * C::PC * C::CPP, NIT = (NCH + C::NT - 1) / C::NT;
    u32x4 pre[NIT];
    auto issue = [&](int tile) {
        const int n = tile / (tiles_y * tiles_x), tr = tile - n * (tiles_y * tiles_x);
        const int iy0 = (tr / tiles_x) * TH * S - 1, ix0 = (tr % tiles_x) * TW * S - 1;
        const unsigned char* gimg = p.g + (size_t)n * p.GH * p.GW * pixg;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * C::NT + tid;
            const int c = idx % C::CPP, pp = idx / C::CPP;
            const int prow = pp / C::PC, pcol = pp - prow * C::PC;
            const int iy = iy0 + prow, ix = ix0 + pcol;
            pre[it] = u32x4{0u, 0u, 0u, 0u};
            if (idx < NCH && (unsigned)iy < (unsigned)p.GH && (unsigned)ix < (unsigned)p.GW)
                pre[it] = *reinterpret_cast<const u32x4*>(gimg + ((size_t)iy * p.GW + ix) * pixg + c * 16);
        }
    };
    if ((int)blockIdx.x < ntiles) issue(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = tile / (tiles_y * tiles_x), tr = tile - n * (tiles_y * tiles_x);
        const int y0 = (tr / tiles_x) * TH, x0 = (tr % tiles_x) * TW;
        __syncthreads();      // the previous tile's store tiles (same LDS) are done; first trip: weights / affine in place after the next one
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * C::NT + tid;
            const int c = idx % C::CPP, pp = idx / C::CPP;
            const int pcol = pp % C::PC;
            if (idx < NCH) *reinterpret_cast<u32x4*>(patch + pp * C::PXB + pswz<C::CPP>(c, pcol) * 16) = pre[it];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);
        // ---- MFMA over the wave's segments ---------------------------------------------------------------------------------
        f32x16 acc[SPW][C::NCB];
#pragma unroll
        for (int s = 0; s < SPW; ++s)
#pragma unroll
            for (int cb = 0; cb < C::NCB; ++cb)
#pragma unroll
                for (int t = 0; t < 16; ++t) acc[s][cb][t] = 0.f;
#pragma unroll
        for (int j = 0; j < C::KS; ++j) {
            // this lane's 8 reduction indices of step j: k0 = j*16 + half*8 -> (tap, first channel)
            const int k0 = j * 16 + half * 8;
            int tap = k0 / CIN;
            const int cch = (k0 - tap * CIN) / 8;
            if (tap > 8) tap = 0;              // K tail of the Cin = 8 layer: the weights there are zero, the data must be finite
            const int kh = tap / 3, kw = tap - kh * 3;
            frag a[C::NCB];
#pragma unroll
            for (int cb = 0; cb < C::NCB; ++cb)
                a[cb] = *reinterpret_cast<const frag*>(wl + (cb * 32 + l31) * C::WROW + j * 32 + half * 16);
#pragma unroll
            for (int s = 0; s < SPW; ++s) {
                const int seg = wave * SPW + s;
                const int sy = seg / (TW / 32), sx = (seg % (TW / 32)) * 32;
                const int prow = sy * S + kh, pcol = (sx + l31) * S + kw;
                const frag b = *reinterpret_cast<const frag*>(patch + (prow * C::PC + pcol) * C::PXB + pswz<C::CPP>(cch, pcol) * 16);
#pragma unroll
                for (int cb = 0; cb < C::NCB; ++cb) acc[s][cb] = Mma32<T>::mma(a[cb], b, acc[s][cb]);
            }
        }
        __syncthreads();      // every wave is done with the patch: its LDS becomes the store tiles
        // ---- epilogue: lane holds D[co = cb*32 + 8g + 4*half + r][pixel l31 of the segment], register 4g + r -----------------
        unsigned char* wst = stage + wave * (32 * C::SROW);
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            const int seg = wave * SPW + s;
            const int oy = y0 + seg / (TW / 32), oxs = x0 + (seg % (TW / 32)) * 32;
            const int nvalid = oy < p.OH ? min(32, max(0, p.OW - oxs)) : 0;       // pixels of this segment inside the image
            // accumulators (eval: after affine + activation) -> the wave's fp32 tile [pixel][channel]
#pragma unroll
            for (int cb = 0; cb < C::NCB; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 h;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[s][cb][4 * g + r];
                        if (affine) {
                            const int co = cb * 32 + 8 * g + 4 * half + r;
                            const float z = v * asc[co] + ash[co];
                            const float zm = mish_f<true>(z), zl = z > 0.f ? z : 0.1f * z;
                            v = p.act == CY_ACT_MISH ? zm : (p.act == CY_ACT_LEAKY ? zl : z);
                        }
                        h[r] = v;
                    }
                    *reinterpret_cast<f32x4*>(wst + l31 * C::SROW + (cb * 32 + 8 * g + 4 * half) * 4) = h;
                }
            // (wave-private tile: the wave's own ds_writes are ordered before its ds_reads by lgkmcnt)
            if (stats) {
                // BatchNorm statistics of the fp32 accumulators: lane c sums column c over the valid pixel rows
#pragma unroll
                for (int cb = 0; cb < (COUT + 63) / 64; ++cb) {
                    const int c = cb * 64 + lane;
                    if (c < COUT) {
                        float sv = 0.f, qv = 0.f;
                        for (int r = 0; r < nvalid; ++r) {
                            const float v = *reinterpret_cast<const float*>(wst + r * C::SROW + c * 4);
                            sv += v;
                            qv += v * v;
                        }
                        ssum[cb] += sv;
                        qsum[cb] += qv;
                    }
                }
            }
            constexpr int CPO = COUT / 8;                 // 16-byte chunks per output pixel
#pragma unroll
            for (int it = 0; it < 32 * CPO / 64; ++it) {
                const int idx = it * 64 + lane, px = idx / CPO, ch = idx - px * CPO;
                if (px >= nvalid || ch * 8 >= p.OC) continue;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(wst + px * C::SROW + ch * 32);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(wst + px * C::SROW + ch * 32 + 16);
                float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const size_t opix = ((size_t)n * p.OH + oy) * p.OW + oxs + px;
                if (affine && p.res) {
                    const tx8 rv = *reinterpret_cast<const tx8*>(reinterpret_cast<const T*>(p.res) + opix * p.ldres + ch * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += (float)rv[e];
                }
                tx8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (T)f[e];
                *reinterpret_cast<tx8*>(p.o + opix * pixo + ch * 16) = v;
            }
        }
    }
    if (stats) {
        // per block: one fp32 atomic per (channel, moment) into one of the CY_STAT_BINS rows
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) {
            const int c = sb * 64 + lane;
            if (c < COUT) {
                red[(wave * 2 + 0) * COUT + c] = ssum[sb];
                red[(wave * 2 + 1) * COUT + c] = qsum[sb];
            }
        }
        __syncthreads();
        float* srow = p.stats + (size_t)(blockIdx.x & (CY_STAT_BINS - 1)) * 2 * p.OC;
        for (int c = tid; c < 2 * COUT; c += C::NT) {
            const int mom = c / COUT, co = c - mom * COUT;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < C::NWAVE; ++w) t += red[(w * 2 + mom) * COUT + co];
            if (co < p.OC) atomicAdd(srow + mom * p.OC + co, t);
        }
    }
}

template <typename T, int CIN, int COUT, int S, int TH, int TW, int SPW>
int direct_launch(const IgemmParams& p, hipStream_t s) {
    typedef DirectCfg<T, CIN, COUT, S, TH, TW, SPW> C;
    static_assert(C::SMEM <= 80 * 1024, "two blocks per CU");
    static unsigned long long attr_done = 0;      // bit d: set for HIP device d
    if (cy_first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&direct3x3_kernel<T, CIN, COUT, S, TH, TW, SPW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
    }
    const long tiles = (long)p.N * ((p.OH + TH - 1) / TH) * ((p.OW + TW - 1) / TW);
    const int slots = 256 * C::BLOCKS_PER_CU;                     // persistent blocks: every CU slot, each looping over tiles
    const unsigned grid = (unsigned)(tiles < slots ? tiles : slots);
    hipLaunchKernelGGL((direct3x3_kernel<T, CIN, COUT, S, TH, TW, SPW>), dim3(grid), dim3(C::NT), C::SMEM, s, p);
    CY_LAUNCH_CHECK();
    return 0;
}

// ---- 1x1 / stride 1 (forward, eval, and the 1x1 dgrad, which is the same GEMM on the dgrad weight matrix) ----------------
// On the 304 / 152 grids these are streams of 100-750 MB through a K of 64 or 128: one K step per tile on the implicit-GEMM
// kernels, i.e. all prologue and epilogue.  Same structure as above without the halo: persistent blocks over tiles of TP
// consecutive pixels, weights [Cout][Cin] resident in LDS, the tile's rows [pixel][Cin] copied once (16-byte chunk index XOR
// the pixel index: conflict-free B reads), the next tile prefetched into registers, epilogue in 64-channel passes.
template <typename T, int CIN, int COUT, int SPW>
struct PwCfg {
    static constexpr int NWAVE = 4, NT = 256, TP = NWAVE * SPW * 32;
    static constexpr int PXB = CIN * 2, CPP = CIN / 8, KS = CIN / 16, NCB = COUT / 32;
    static constexpr int WROW = CIN * 2 + 16;
    static constexpr int W_BYTES = COUT * WROW;
    static constexpr int PATCH_BYTES = TP * PXB;
    static constexpr int CW = COUT < 64 ? COUT : 64;                        // channels per epilogue pass
    static constexpr int SROW = CW * 4 + 16;
    static constexpr int ST_BYTES = NWAVE * 32 * SROW;
    static constexpr int P_BYTES = PATCH_BYTES > ST_BYTES ? PATCH_BYTES : ST_BYTES;
    static constexpr int RED_BYTES = NWAVE * 2 * COUT * 4 + 2 * COUT * 4;
    static constexpr int SMEM = W_BYTES + P_BYTES + RED_BYTES;
    static constexpr int NCH = TP * CPP, NIT = NCH / NT;
    static_assert(CIN % 16 == 0 && COUT % 32 == 0 && NCH % NT == 0 && SMEM <= 80 * 1024, "tile layout");
};

// EXTRA: the input-gradient variants that read while they store -- gradient fan-in (CY_CONV_ACCUM) and / or the BatchNorm-backward
// sums of the producer layer (CY_CONV_BNBWD_SUMS); a separate instantiation so that the forward / eval kernels keep their registers
// BNIN (cy_conv1x1_bn_in, round 6: the measured consumer-side BatchNorm): the rows this kernel reads are a PRE-BatchNorm tensor;
// on their way from the registers into LDS they become act(x * in_scale[c] + in_shift[c]) (p.bn_gamma / p.bn_beta = the producer
// layer's folded scale / shift per INPUT channel, p.act its activation) and are also written to p.o2 -- the activated tensor the
// weight gradient of this conv and the backward pass still need.  Every pixel tile is staged by exactly one block, so that side
// output is written once; the producer's separate BatchNorm + activation pass (one read of the pre-BN tensor, one launch) goes.
template <typename T, int CIN, int COUT, int SPW, bool EXTRA, bool BNIN = false>
__global__ void __launch_bounds__(256, 2) direct1x1_kernel(const IgemmParams p) {
    typedef PwCfg<T, CIN, COUT, SPW> C;
    typedef typename Mma32<T>::frag frag;
    typedef T tx8 __attribute__((ext_vector_type(8)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wl = smem;
    unsigned char* patch = smem + C::W_BYTES;
    unsigned char* stage = patch;
    float* red = reinterpret_cast<float*>(patch + C::P_BYTES);
    float* asc = red + C::NWAVE * 2 * COUT;
    float* ash = asc + COUT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    {
        constexpr int CPRW = CIN * 2 / 16;
        for (int idx = tid; idx < COUT * CPRW; idx += C::NT) {
            const int row = idx / CPRW, ch = idx - row * CPRW;
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (row < p.wrows) v = *reinterpret_cast<const u32x4*>(p.w + (size_t)row * (CIN * 2) + ch * 16);
            *reinterpret_cast<u32x4*>(wl + row * C::WROW + ch * 16) = v;
        }
    }
    const bool stats = (p.flags & CY_CONV_STATS) != 0, affine = (p.flags & CY_CONV_AFFINE_ACT) != 0;
    const bool accum = EXTRA && (p.flags & CY_CONV_ACCUM) != 0, bnsum = EXTRA && (p.flags & CY_CONV_BNBWD_SUMS) != 0;
    // store phase: lane -> (pixel lane / CPO + 64 / CPO * it, 16-byte chunk lane % CPO of the pass): the chunk is a lane constant
    constexpr int CPO = C::CW / 8, NPASS = COUT / C::CW, NSI = 32 * CPO / 64;
    const int cl8 = (lane % CPO) * 8;
    // CY_CONV_BNBWD_SUMS (1x1 dgrad only): BatchNorm-backward sums of the layer whose output gradient this launch stores, as in
    // conv_pipe.hip -- res / ldres = that layer's pre-BN tensor, aff_scale / aff_shift its affine, bn_mean / bn_invstd its statistics
    // (its affine sits in LDS, where the eval epilogue keeps the layer's own: asc / ash)
    float s1[EXTRA ? NPASS : 1][8], s2[EXTRA ? NPASS : 1][8];
#pragma unroll
    for (int q = 0; q < (EXTRA ? NPASS : 1); ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[q][e] = 0.f; s2[q][e] = 0.f; }
    if (affine || (p.flags & CY_CONV_BNBWD_SUMS))
        for (int c = tid; c < COUT; c += C::NT) { asc[c] = p.aff_scale[min(c, p.OC - 1)]; ash[c] = p.aff_shift[min(c, p.OC - 1)]; }
    constexpr int NSB = (COUT + 63) / 64;
    float ssum[NSB], qsum[NSB];
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) { ssum[sb] = 0.f; qsum[sb] = 0.f; }
    const long M = p.M;
    const int ntiles = (int)((M + C::TP - 1) / C::TP);
    const size_t pixg = (size_t)p.ldg * 2, pixo = (size_t)p.ldo * 2;
    // BNIN: this thread's 16-byte chunk index tid % CPP is the same for every row it stages (NT % CPP == 0)
    static_assert(C::NT % C::CPP == 0, "a staging thread keeps its channel chunk");
    float isc[BNIN ? 8 : 1], ish[BNIN ? 8 : 1];
    if constexpr (BNIN) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            isc[e] = p.bn_gamma[(tid % C::CPP) * 8 + e];
            ish[e] = p.bn_beta[(tid % C::CPP) * 8 + e];
        }
    }

    u32x4 pre[C::NIT];
    auto issue = [&](int tile) {
        const long m0 = (long)tile * C::TP;
#pragma unroll
        for (int it = 0; it < C::NIT; ++it) {
            const int idx = it * C::NT + tid, px = idx / C::CPP, c = idx - px * C::CPP;
            pre[it] = u32x4{0u, 0u, 0u, 0u};
            if (m0 + px < M) pre[it] = *reinterpret_cast<const u32x4*>(p.g + (size_t)(m0 + px) * pixg + c * 16);
        }
    };
    // (the 128-channel EXTRA instantiation has no registers to spare for the next tile's rows: it fetches each tile when it
    // gets there and leaves the latency to the second block of the CU)
    constexpr bool PRE = !(EXTRA && COUT > 64);
    if (PRE && (int)blockIdx.x < ntiles) issue(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long m0 = (long)tile * C::TP;
        if (!PRE) issue(tile);
        __syncthreads();
#pragma unroll
        for (int it = 0; it < C::NIT; ++it) {
            const int idx = it * C::NT + tid, px = idx / C::CPP, c = idx - px * C::CPP;
            if constexpr (BNIN) {
                float f[8];
                chunk_to_f32<T>(pre[it], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float z = f[e] * isc[e] + ish[e];
                    const float zm = mish_f<true>(z), zl = z > 0.f ? z : 0.1f * z;
                    f[e] = p.act == CY_ACT_MISH ? zm : (p.act == CY_ACT_LEAKY ? zl : z);
                }
                pre[it] = f32_to_chunk<T>(f);
                if (m0 + px < M) *reinterpret_cast<u32x4*>(p.o2 + (size_t)(m0 + px) * ((size_t)p.ldo2 * 2) + c * 16) = pre[it];
            }
            *reinterpret_cast<u32x4*>(patch + px * C::PXB + ((c ^ (px & (C::CPP - 1))) << 4)) = pre[it];
        }
        __syncthreads();
        if (PRE && tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);
        // epilogue operands (pre-BN tensor for the sums, stored gradient for a fan-in launch) of one (segment, channel pass) at a
        // time: the first before the MFMAs, each next one while the previous is being stored
        tx8 rawv[EXTRA ? NSI : 1], oldv[EXTRA ? NSI : 1];
        auto prefetch = [&](int sq) {
            const int s = sq / NPASS, q = sq - s * NPASS;
#pragma unroll
            for (int it = 0; it < NSI; ++it) {
                const long m = m0 + (wave * SPW + s) * 32 + it * (64 / CPO) + lane / CPO;
                const int co = q * C::CW + cl8;
                const bool ok = m < M && co < p.OC;
                rawv[it] = tx8{};
                oldv[it] = tx8{};
                if (bnsum && ok) rawv[it] = *reinterpret_cast<const tx8*>(reinterpret_cast<const T*>(p.res) + (size_t)m * p.ldres + co);
                if (accum && ok) oldv[it] = *reinterpret_cast<const tx8*>(reinterpret_cast<const T*>(p.o + (size_t)m * pixo) + co);
            }
        };
        if constexpr (EXTRA) { if (bnsum || accum) prefetch(0); }
        f32x16 acc[SPW][C::NCB];
#pragma unroll
        for (int s = 0; s < SPW; ++s)
#pragma unroll
            for (int cb = 0; cb < C::NCB; ++cb)
#pragma unroll
                for (int t = 0; t < 16; ++t) acc[s][cb][t] = 0.f;
#pragma unroll
        for (int j = 0; j < C::KS; ++j) {
            const int cch = 2 * j + half;
            if constexpr (EXTRA) asm volatile("" ::: "memory");    // one K step's fragments at a time: this variant is short of registers
            frag a[C::NCB];
#pragma unroll
            for (int cb = 0; cb < C::NCB; ++cb)
                a[cb] = *reinterpret_cast<const frag*>(wl + (cb * 32 + l31) * C::WROW + j * 32 + half * 16);
#pragma unroll
            for (int s = 0; s < SPW; ++s) {
                const int px = (wave * SPW + s) * 32 + l31;
                const frag b = *reinterpret_cast<const frag*>(patch + px * C::PXB + ((cch ^ (px & (C::CPP - 1))) << 4));
#pragma unroll
                for (int cb = 0; cb < C::NCB; ++cb) acc[s][cb] = Mma32<T>::mma(a[cb], b, acc[s][cb]);
            }
        }
        __syncthreads();
        unsigned char* wst = stage + wave * (32 * C::SROW);
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            const long ms = m0 + (wave * SPW + s) * 32;
            const int nvalid = (int)(M - ms < 32 ? (M - ms > 0 ? M - ms : 0) : 32);
#pragma unroll
            for (int pass = 0; pass < COUT / C::CW; ++pass) {      // 64 channels (two 32-channel MFMA blocks) at a time
#pragma unroll
                for (int cbl = 0; cbl < C::CW / 32; ++cbl) {
                    const int cb = pass * (C::CW / 32) + cbl;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 h;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = acc[s][cb][4 * g + r];
                            if (affine) {
                                const int co = cb * 32 + 8 * g + 4 * half + r;
                                const float z = v * asc[co] + ash[co];
                                const float zm = mish_f<true>(z), zl = z > 0.f ? z : 0.1f * z;
                                v = p.act == CY_ACT_MISH ? zm : (p.act == CY_ACT_LEAKY ? zl : z);
                            }
                            h[r] = v;
                        }
                        *reinterpret_cast<f32x4*>(wst + l31 * C::SROW + (cbl * 32 + 8 * g + 4 * half) * 4) = h;
                    }
                }
                if (stats && lane < C::CW) {
                    float sv = 0.f, qv = 0.f;
                    for (int r = 0; r < nvalid; ++r) {
                        const float v = *reinterpret_cast<const float*>(wst + r * C::SROW + lane * 4);
                        sv += v;
                        qv += v * v;
                    }
                    // lane c of pass `pass` owns channel pass*CW + c: NSB accumulators indexed by the pass (CW = 64) or 0
                    ssum[C::CW == 64 ? pass : 0] += sv;
                    qsum[C::CW == 64 ? pass : 0] += qv;
                }
#pragma unroll
                for (int it = 0; it < NSI; ++it) {
                    const int idx = it * 64 + lane, px = idx / CPO, ch = idx - px * CPO;
                    const int co = pass * C::CW + ch * 8;
                    if (px >= nvalid || co >= p.OC) continue;
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(wst + px * C::SROW + ch * 32);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(wst + px * C::SROW + ch * 32 + 16);
                    float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    const size_t opix = (size_t)(ms + px);
                    T* dst = reinterpret_cast<T*>(p.o + opix * pixo) + co;
                    if (affine && p.res) {
                        const tx8 rv = *reinterpret_cast<const tx8*>(reinterpret_cast<const T*>(p.res) + opix * p.ldres + co);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += (float)rv[e];
                    }
                    if constexpr (EXTRA) {
                        if (accum) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] += (float)oldv[it][e];
                        }
                    }
                    tx8 v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (T)f[e];
                    *reinterpret_cast<tx8*>(dst) = v;
                    if constexpr (EXTRA) if (bnsum) {
                        const f32x4 c0 = *reinterpret_cast<const f32x4*>(asc + co), c1 = *reinterpret_cast<const f32x4*>(asc + co + 4);
                        const f32x4 h0 = *reinterpret_cast<const f32x4*>(ash + co), h1 = *reinterpret_cast<const f32x4*>(ash + co + 4);
                        const float bsc[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                        const float bsh[8] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float x = (float)rawv[it][e];
                            const float z = x * bsc[e] + bsh[e];
                            const float dm = mish_grad<true>(z), dl = z > 0.f ? 1.f : 0.1f;
                            const float dz = (float)v[e] * (p.act == CY_ACT_MISH ? dm : (p.act == CY_ACT_LEAKY ? dl : 1.f));
                            s1[pass][e] += dz;
                            s2[pass][e] += dz * x;
                        }
                    }
                }
                if constexpr (EXTRA) {
                    if ((bnsum || accum) && s * NPASS + pass + 1 < SPW * NPASS) prefetch(s * NPASS + pass + 1);
                }
            }
        }
    }
    if (bnsum) {
        // lanes with equal lane % CPO hold the same channels: fold them, the waves through LDS, one atomic per (channel, moment)
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NPASS; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int m = CPO; m < 64; m <<= 1) {
                    s1[q][e] += __shfl_xor(s1[q][e], m);
                    s2[q][e] += __shfl_xor(s2[q][e], m);
                }
            }
        if (lane < CPO) {
#pragma unroll
            for (int q = 0; q < NPASS; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = q * C::CW + cl8 + e, ch = min(c, p.OC - 1);
                    red[(wave * 2 + 0) * COUT + c] = s1[q][e];
                    red[(wave * 2 + 1) * COUT + c] = (s2[q][e] - p.bn_mean[ch] * s1[q][e]) * p.bn_invstd[ch];
                }
        }
        __syncthreads();
        float* srow = p.stats + (size_t)(blockIdx.x & (CY_STAT_BINS - 1)) * 2 * p.OC;
        for (int c = tid; c < 2 * COUT; c += C::NT) {
            const int mom = c / COUT, co = c - mom * COUT;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < C::NWAVE; ++w) t += red[(w * 2 + mom) * COUT + co];
            if (co < p.OC) atomicAdd(srow + mom * p.OC + co, t);
        }
    }
    if (stats) {
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) {
            const int c = sb * 64 + lane;
            if (lane < C::CW && c < COUT) {
                red[(wave * 2 + 0) * COUT + c] = ssum[sb];
                red[(wave * 2 + 1) * COUT + c] = qsum[sb];
            }
        }
        __syncthreads();
        float* srow = p.stats + (size_t)(blockIdx.x & (CY_STAT_BINS - 1)) * 2 * p.OC;
        for (int c = tid; c < 2 * COUT; c += C::NT) {
            const int mom = c / COUT, co = c - mom * COUT;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < C::NWAVE; ++w) t += red[(w * 2 + mom) * COUT + co];
            if (co < p.OC) atomicAdd(srow + mom * p.OC + co, t);
        }
    }
}

template <typename T, int CIN, int COUT, int SPW, bool EXTRA, bool BNIN = false>
int pw_launch_x(const IgemmParams& p, hipStream_t s) {
    typedef PwCfg<T, CIN, COUT, SPW> C;
    static unsigned long long attr_done = 0;      // bit d: set for HIP device d
    if (cy_first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&direct1x1_kernel<T, CIN, COUT, SPW, EXTRA, BNIN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
    }
    const long tiles = (p.M + C::TP - 1) / C::TP;
    const unsigned grid = (unsigned)(tiles < 512 ? tiles : 512);
    hipLaunchKernelGGL((direct1x1_kernel<T, CIN, COUT, SPW, EXTRA, BNIN>), dim3(grid), dim3(256), C::SMEM, s, p);
    CY_LAUNCH_CHECK();
    return 0;
}

template <typename T, int CIN, int COUT, int SPW>
int pw_launch(const IgemmParams& p, hipStream_t s) {
    // (the EXTRA instantiations run one 32-pixel segment per wave: half the accumulators, room for the prefetched operands --
    // no kernel of the train step may spill to scratch memory, see build.py)
    if (p.flags & (CY_CONV_ACCUM | CY_CONV_BNBWD_SUMS)) return pw_launch_x<T, CIN, COUT, 1, true>(p, s);
    return pw_launch_x<T, CIN, COUT, SPW, false>(p, s);
}


// ---- 3x3 / stride 2 / pad 1 input gradient, small channel counts (dX 32 <- dY 64 @304 -> 608) ------------------------------
// The four parity classes of a stride-2 dgrad are stride-1 "convolutions" of dY with 1, 2, 2 and 4 taps whose outputs
// interleave on the dX lattice.  On the implicit-GEMM kernels the 32 <- 64 layer is one to four K steps per tile -- all
// prologue, gather table and epilogue: 315 us for 567 MB (the slowest launch of the train step) against an HBM floor of
// ~115 us.  Here, like the forward direct kernel: persistent blocks over tiles of 8 x 64 dX pixels, the dgrad weights
// [OC][9 * GC] resident in LDS, the (4 + 1) x (32 + 1) dY patch read once (next tile prefetched into registers), wave w owns
// dY row w of the tile = dX rows 2w, 2w + 1: per dX row the two column classes are computed (32-pixel segments, K = taps * GC)
// and interleaved in a wave-private fp32 LDS tile, so the row leaves as 64 consecutive pixels, 16 B per lane.  Optional
// epilogues: gradient fan-in (read-add-store) and the BatchNorm-backward sums of the layer whose output gradient this is
// (CY_CONV_BNBWD_SUMS, as in conv_pipe.hip: one read of that layer's pre-BN tensor instead of a separate reduce pass).
template <typename T, int GC, int OC>
struct S2Cfg {
    static constexpr int THO = 8, TWO = 64, NWAVE = THO / 2, NT = NWAVE * 64;
    static constexpr int PR = THO / 2 + 1, PC = TWO / 2 + 1;
    static constexpr int PXB = GC * 2, CPP = GC / 8;
    static constexpr int K = 9 * GC, WROW = K * 2 + 16, NCB = OC / 32;
    static constexpr int W_BYTES = OC * WROW;
    static constexpr int SROW = OC * 4 + 16;
    static constexpr int ST_BYTES = NWAVE * TWO * SROW;
    static constexpr int PATCH_BYTES = (PR * PC * PXB + 15) / 16 * 16;
    static constexpr int P_BYTES = PATCH_BYTES > ST_BYTES ? PATCH_BYTES : ST_BYTES;
    static constexpr int RED_BYTES = NWAVE * 2 * OC * 4;
    static constexpr int SMEM = W_BYTES + P_BYTES + RED_BYTES;
    static constexpr int NCH = PR * PC * CPP, NIT = (NCH + NT - 1) / NT;
    static_assert((CPP == 8 || CPP == 16) && OC % 32 == 0 && SMEM <= 80 * 1024, "tile layout");
};

// physical 16-byte chunk of logical chunk c of patch column pcol: 16 consecutive columns x one logical chunk = 16 distinct
// bank groups (128-byte pixels alternate bank halves, 256-byte pixels all start on bank 0)
template <int CPP>
__device__ __forceinline__ int s2swz(int c, int pcol) {
    return CPP == 8 ? (c ^ ((pcol >> 1) & 7)) : (c ^ (pcol & 15));
}

template <typename T, int GC, int OC>
__global__ void __launch_bounds__(256, 2) direct_s2dgrad_kernel(const IgemmParams p) {
    typedef S2Cfg<T, GC, OC> C;
    typedef typename Mma32<T>::frag frag;
    typedef T tx8 __attribute__((ext_vector_type(8)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wl = smem;
    unsigned char* patch = smem + C::W_BYTES;
    unsigned char* stage = patch;
    float* red = reinterpret_cast<float*>(patch + C::P_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    {
        constexpr int CPRW = C::K * 2 / 16;
        for (int idx = tid; idx < OC * (CPRW + 1); idx += C::NT) {
            const int row = idx / (CPRW + 1), ch = idx - row * (CPRW + 1);
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (ch < CPRW && row < p.wrows) v = *reinterpret_cast<const u32x4*>(p.w + (size_t)row * (C::K * 2) + ch * 16);
            *reinterpret_cast<u32x4*>(wl + row * C::WROW + ch * 16) = v;
        }
    }
    const bool accum = (p.flags & CY_CONV_ACCUM) != 0, bnsum = (p.flags & CY_CONV_BNBWD_SUMS) != 0;
    const int tiles_x = (p.OW + C::TWO - 1) / C::TWO, tiles_y = (p.OH + C::THO - 1) / C::THO;
    const int ntiles = p.N * tiles_y * tiles_x;
    const size_t pixg = (size_t)p.ldg * 2, pixo = (size_t)p.ldo * 2;
    // store phase: lane -> (pixel lane / CPO + 64 / CPO * it, 16-byte chunk lane % CPO): the chunk is a lane constant
    constexpr int CPO = OC / 8, PPI = 64 / CPO;
    const int cl8 = (lane % CPO) * 8;
    float bsc[8], bsh[8], s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bsc[e] = 0.f; bsh[e] = 0.f; s1[e] = 0.f; s2[e] = 0.f; }
    if (bnsum) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = min(cl8 + e, p.OC - 1);
            bsc[e] = p.aff_scale[ch]; bsh[e] = p.aff_shift[ch];
        }
    }

    u32x4 pre[C::NIT];
    auto issue = [&](int tile) {
        const int n = tile / (tiles_y * tiles_x), tr = tile - n * (tiles_y * tiles_x);
        const int a0 = (tr / tiles_x) * (C::THO / 2), b0 = (tr % tiles_x) * (C::TWO / 2);
        const unsigned char* gimg = p.g + (size_t)n * p.GH * p.GW * pixg;
#pragma unroll
        for (int it = 0; it < C::NIT; ++it) {
            const int idx = it * C::NT + tid;
            const int c = idx % C::CPP, pp = idx / C::CPP;
            const int prow = pp / C::PC, pcol = pp - prow * C::PC;
            const int iy = a0 + prow, ix = b0 + pcol;
            pre[it] = u32x4{0u, 0u, 0u, 0u};
            if (idx < C::NCH && iy < p.GH && ix < p.GW)
                pre[it] = *reinterpret_cast<const u32x4*>(gimg + ((size_t)iy * p.GW + ix) * pixg + c * 16);
        }
    };
    if ((int)blockIdx.x < ntiles) issue(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = tile / (tiles_y * tiles_x), tr = tile - n * (tiles_y * tiles_x);
        const int y0 = (tr / tiles_x) * C::THO, x0 = (tr % tiles_x) * C::TWO;
        __syncthreads();      // the previous tile's store tiles (same LDS) are done; first trip: weights in place after the next one
#pragma unroll
        for (int it = 0; it < C::NIT; ++it) {
            const int idx = it * C::NT + tid;
            const int c = idx % C::CPP, pp = idx / C::CPP;
            const int pcol = pp % C::PC;
            if (idx < C::NCH) *reinterpret_cast<u32x4*>(patch + pp * C::PXB + s2swz<C::CPP>(c, pcol) * 16) = pre[it];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);
        // epilogue operands of this wave's dX rows (the producer layer's pre-BN tensor for the sums, the gradient already
        // stored for a fan-in launch): the first row's are requested NOW, so that their latency hides behind the MFMAs; the
        // second row's once the first row's are consumed
        constexpr int NSI = C::TWO / PPI;          // store instructions per dX row
        tx8 rawv[NSI], oldv[NSI];
        auto prefetch = [&](int ph) {
            const int oy = y0 + 2 * wave + ph;
#pragma unroll
            for (int it = 0; it < NSI; ++it) {
                const int px = it * PPI + lane / CPO;
                const bool ok = oy < p.OH && x0 + px < p.OW && cl8 < p.OC;
                const size_t opix = ((size_t)n * p.OH + oy) * p.OW + x0 + px;
                rawv[it] = tx8{};
                oldv[it] = tx8{};
                if (bnsum && ok) rawv[it] = *reinterpret_cast<const tx8*>(reinterpret_cast<const T*>(p.res) + opix * p.ldres + cl8);
                if (accum && ok) oldv[it] = *reinterpret_cast<const tx8*>(reinterpret_cast<const T*>(p.o + opix * pixo) + cl8);
            }
        };
        if (bnsum || accum) prefetch(0);
        // ---- MFMA: acc[ph][pw][cb] = class (ph, pw) of dY row `wave`: 32 dX pixels (x = x0 + 2 l31 + pw) x 32 channels ------
        f32x16 acc[2][2][C::NCB];
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
            for (int pw = 0; pw < 2; ++pw)
#pragma unroll
                for (int cb = 0; cb < C::NCB; ++cb)
#pragma unroll
                    for (int t = 0; t < 16; ++t) acc[ph][pw][cb][t] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ph = (kh & 1) ^ 1;                    // taps with (ph + 1 - kh) even
            const int dh = (ph + 1 - kh) >> 1;              // dY row of dX row 2a + ph under tap kh: a + dh
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int pw = (kw & 1) ^ 1, dw = (pw + 1 - kw) >> 1;
                const int pcol = l31 + dw;
                const unsigned char* prow = patch + ((wave + dh) * C::PC + pcol) * C::PXB;
                asm volatile("" ::: "memory");      // one tap's fragments at a time (hipcc otherwise hoists all 72 reads: spills)
#pragma unroll
                for (int j = 0; j < GC / 16; ++j) {
                    const frag b = *reinterpret_cast<const frag*>(prow + s2swz<C::CPP>(2 * j + half, pcol) * 16);
#pragma unroll
                    for (int cb = 0; cb < C::NCB; ++cb) {
                        const frag a = *reinterpret_cast<const frag*>(wl + (cb * 32 + l31) * C::WROW + ((kh * 3 + kw) * GC + j * 16) * 2 + half * 16);
                        acc[ph][pw][cb] = Mma32<T>::mma(a, b, acc[ph][pw][cb]);
                    }
                }
            }
        }
        __syncthreads();      // every wave is done with the patch: its LDS becomes the store tiles
        unsigned char* wst = stage + wave * (C::TWO * C::SROW);
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const int oy = y0 + 2 * wave + ph;
            // both column classes of the row -> [64 pixels][OC] fp32, pixel 2 l31 + pw
#pragma unroll
            for (int pw = 0; pw < 2; ++pw)
#pragma unroll
                for (int cb = 0; cb < C::NCB; ++cb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 h;
#pragma unroll
                        for (int r = 0; r < 4; ++r) h[r] = acc[ph][pw][cb][4 * g + r];
                        *reinterpret_cast<f32x4*>(wst + (2 * l31 + pw) * C::SROW + (cb * 32 + 8 * g + 4 * half) * 4) = h;
                    }
            // (wave-private tile: the wave's own ds_writes are ordered before its ds_reads by lgkmcnt)
            const int nvalid = oy < p.OH ? min(C::TWO, max(0, p.OW - x0)) : 0;
#pragma unroll
            for (int it = 0; it < NSI; ++it) {
                const int px = it * PPI + lane / CPO;
                if (px >= nvalid || cl8 >= p.OC) continue;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(wst + px * C::SROW + cl8 * 4);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(wst + px * C::SROW + cl8 * 4 + 16);
                float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const size_t opix = ((size_t)n * p.OH + oy) * p.OW + x0 + px;
                if (accum) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += (float)oldv[it][e];
                }
                tx8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (T)f[e];
                *reinterpret_cast<tx8*>(reinterpret_cast<T*>(p.o + opix * pixo) + cl8) = v;
                if (bnsum) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float x = (float)rawv[it][e];
                        const float z = x * bsc[e] + bsh[e];
                        const float dm = mish_grad<true>(z), dl = z > 0.f ? 1.f : 0.1f;
                        const float dz = (float)v[e] * (p.act == CY_ACT_MISH ? dm : (p.act == CY_ACT_LEAKY ? dl : 1.f));
                        s1[e] += dz;
                        s2[e] += dz * x;
                    }
                }
            }
            if (ph == 0 && (bnsum || accum)) prefetch(1);
        }
    }
    if (bnsum) {
        // lanes with equal lane % CPO hold the same 8 channels: fold them, then the waves through LDS, then one atomic per
        // (channel, moment) of the block into one of the CY_STAT_BINS rows -- the table layout of the forward statistics
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int m = CPO; m < 64; m <<= 1) {
                s1[e] += __shfl_xor(s1[e], m);
                s2[e] += __shfl_xor(s2[e], m);
            }
        }
        if (lane < CPO) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ch = min(cl8 + e, p.OC - 1);
                red[(wave * 2 + 0) * OC + cl8 + e] = s1[e];
                red[(wave * 2 + 1) * OC + cl8 + e] = (s2[e] - p.bn_mean[ch] * s1[e]) * p.bn_invstd[ch];
            }
        }
        __syncthreads();
        float* srow = p.stats + (size_t)(blockIdx.x & (CY_STAT_BINS - 1)) * 2 * p.OC;
        for (int c = tid; c < 2 * OC; c += C::NT) {
            const int mom = c / OC, co = c - mom * OC;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < C::NWAVE; ++w) t += red[(w * 2 + mom) * OC + co];
            if (co < p.OC) atomicAdd(srow + mom * p.OC + co, t);
        }
    }
}

template <typename T, int GC, int OC>
int s2dgrad_launch(const IgemmParams& p, hipStream_t s) {
    typedef S2Cfg<T, GC, OC> C;
    static unsigned long long attr_done = 0;      // bit d: set for HIP device d
    if (cy_first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&direct_s2dgrad_kernel<T, GC, OC>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
    }
    const long tiles = (long)p.N * ((p.OH + C::THO - 1) / C::THO) * ((p.OW + C::TWO - 1) / C::TWO);
    const unsigned grid = (unsigned)(tiles < 512 ? tiles : 512);
    hipLaunchKernelGGL((direct_s2dgrad_kernel<T, GC, OC>), dim3(grid), dim3(C::NT), C::SMEM, s, p);
    CY_LAUNCH_CHECK();
    return 0;
}

template <typename T>
int s2dgrad_dispatch(const IgemmParams& p, hipStream_t s, int* used) {
    *used = 1;
    if (p.GC == 64 && p.OC == 32) return s2dgrad_launch<T, 64, 32>(p, s);
    *used = 0;
    return 0;
}

template <typename T>
int pw_dispatch(const IgemmParams& p, hipStream_t s, int* used) {
    *used = 1;
#define CY_PW(CI, CO, SP) if (p.GC == CI && p.OC == CO) return pw_launch<T, CI, CO, SP>(p, s);
    CY_PW(64, 64, 2) CY_PW(128, 64, 1) CY_PW(64, 128, 1) CY_PW(64, 32, 2) CY_PW(32, 64, 2)
#undef CY_PW
    *used = 0;
    return 0;
}

template <typename T>
int direct_dispatch(const IgemmParams& p, hipStream_t s, int* used) {
    *used = 1;
    if (p.GC == 8 && p.OC == 32 && p.stride == 1) return direct_launch<T, 8, 32, 1, 4, 64, 2>(p, s);
    if (p.GC == 32 && p.OC == 64 && p.stride == 1) return direct_launch<T, 32, 64, 1, 4, 64, 2>(p, s);
    if (p.GC == 32 && p.OC == 64 && p.stride == 2) return direct_launch<T, 32, 64, 2, 4, 32, 1>(p, s);
    *used = 0;
    return 0;
}

}  // namespace

static int g_direct_mode = -1;
static int64_t g_direct_launches = 0;

extern "C" int64_t cy_direct_launches(void) { return g_direct_launches; }

// Which launches take the direct kernel: 16-bit forward 3x3 convs with pad 1 of the three instantiated (Cin, Cout, stride)
// shapes, whole-tensor views (every (Cin, Cout) chunk aligned), BN statistics into shared bins or the eval-mode epilogue.
// CY_CONV_TILE(1) keeps a call on the 4-wave kernels (A/B and the autotuner's baseline); CY_CONV_DIRECT=0 switches it off.
int cy_direct_try(const cyk::IgemmParams& p, int dtype, hipStream_t s, int* used) {
    *used = 0;
    if (g_direct_mode < 0) {
        const char* e = getenv("CY_CONV_DIRECT");
        g_direct_mode = e ? atoi(e) : 1;
    }
    const int hint = (p.flags >> CY_CONV_TILE_SHIFT) & 15;
    if (!g_direct_mode || hint == 1 || (dtype != CY_F16 && dtype != CY_BF16)) return 0;
    if (p.stat_det || (p.flags & CY_CONV_BIAS_F32OUT)) return 0;
    if (p.ldg % 8 || p.ldo % 8 || ((uintptr_t)p.g & 15) || ((uintptr_t)p.o & 15) || ((uintptr_t)p.w & 15)) return 0;
    if (p.res && (p.ldres % 8 || ((uintptr_t)p.res & 15))) return 0;
    if (p.transposed && p.stride == 2 && p.ks == 3 && p.pad == 1 && p.ncls == 4) {
        // the whole stride-2 input gradient (all four parity classes) of the small-channel layers
        // (taken without a hint or with hint 10; the capacity hints 2-9 name the pipelined kernel)
        if (hint != 0 && hint != 10) return 0;
        if ((p.flags & (CY_CONV_STATS | CY_CONV_AFFINE_ACT)) || p.OH != 2 * p.GH || p.OW != 2 * p.GW || p.OC % 8) return 0;
        const int rc = dtype == CY_F16 ? s2dgrad_dispatch<f16>(p, s, used) : s2dgrad_dispatch<bf16>(p, s, used);
        if (rc == 0 && *used) ++g_direct_launches;
        return rc;
    }
    if (p.ks == 1 && p.stride == 1 && p.pad == 0) {
        // 1x1 streams: worth it where the launch is long enough to fill the persistent grid several times over (the 152 / 304
        // grids at batch 16); smaller launches stay with the implicit-GEMM kernels unless the caller asks (hint 10)
        if (p.M < 256L * 1024 && hint != 10) return 0;
        if (p.OH != p.GH || p.OW != p.GW) return 0;
        if ((p.flags & CY_CONV_BNBWD_SUMS) && (p.flags & (CY_CONV_STATS | CY_CONV_AFFINE_ACT))) return 0;
        const int rc = dtype == CY_F16 ? pw_dispatch<f16>(p, s, used) : pw_dispatch<bf16>(p, s, used);
        if (rc == 0 && *used) ++g_direct_launches;
        return rc;
    }
    if (p.flags & CY_CONV_BNBWD_SUMS) return 0;
    if (p.ks != 3 || p.pad != 1 || p.transposed || (p.flags & CY_CONV_ACCUM)) return 0;
    if (p.OH != (p.GH + 2 - 3) / p.stride + 1 || p.OW != (p.GW + 2 - 3) / p.stride + 1) return 0;
    const int rc = dtype == CY_F16 ? direct_dispatch<f16>(p, s, used) : direct_dispatch<bf16>(p, s, used);
    if (rc == 0 && *used) ++g_direct_launches;
    return rc;
}

// The consumer-side BatchNorm prototype VERDICT r5 #6 asked to be MEASURED (tools/bn_in_micro.py, profiles/r06_consumer_side_bn.txt):
// out = act_in(x * in_scale + in_shift) (*) W as ONE launch of the 1x1 streaming kernel, the activated rows written to `act_out`
// on the way (the weight gradient of this conv reads them), BatchNorm statistics of `out` into the shared bins as cy_conv_igemm
// does with CY_CONV_STATS.  x: [M][Cin] pre-BN rows of the producer layer (16-bit), in_scale / in_shift: its folded affine
// (cy_bn_finalize's scale / shift).  Instantiated for the two shapes of the 304 / 152 grids that stay on this kernel:
// 64 -> 128 (the stage-1 sibling pair as one conv) and 64 -> 64.  CY_ERR_UNSUPPORTED otherwise.
extern "C" int cy_conv1x1_bn_in(const void* x, int64_t M, int Cin, int ldx, const float* in_scale, const float* in_shift, int act_in,
                                void* act_out, int ld_act, const void* w, int wrows, void* out, int OC, int ldo, int dtype,
                                int flags, float* stats_part, cy_stream_t s) {
    CY_ENTER();
    if (!x || !in_scale || !in_shift || !act_out || !w || !out || M <= 0 || M > 0x7FFFFFFF) return CY_ERR_ARG;
    if (flags & ~CY_CONV_STATS) return CY_ERR_ARG;
    if ((flags & CY_CONV_STATS) && !stats_part) return CY_ERR_ARG;
    if (dtype != CY_F16 && dtype != CY_BF16) return CY_ERR_UNSUPPORTED;
    if (ldx % 8 || ldo % 8 || ld_act % 8 || ((uintptr_t)x & 15) || ((uintptr_t)out & 15) || ((uintptr_t)act_out & 15) || ((uintptr_t)w & 15))
        return CY_ERR_ARG;
    cyk::IgemmParams p;
    memset(&p, 0, sizeof(p));
    p.g = (const unsigned char*)x; p.w = (const unsigned char*)w; p.o = (unsigned char*)out;
    p.o2 = (unsigned char*)act_out; p.ldo2 = ld_act;
    p.bn_gamma = in_scale; p.bn_beta = in_shift; p.act = act_in;
    p.stats = stats_part; p.flags = flags;
    p.N = 1; p.GH = p.OH = 1; p.GW = p.OW = (int)M; p.GC = Cin; p.ldg = ldx; p.OC = OC; p.ldo = ldo;
    p.ks = 1; p.stride = 1; p.pad = 0; p.K = Cin; p.M = (int)M; p.wrows = wrows;
#define CY_BNIN(CI, CO) \
    if (Cin == CI && OC == CO)                                                                                     \
        return dtype == CY_F16 ? pw_launch_x<f16, CI, CO, 1, false, true>(p, cy_s(s)) : pw_launch_x<bf16, CI, CO, 1, false, true>(p, cy_s(s));
    CY_BNIN(64, 128) CY_BNIN(64, 64)
#undef CY_BNIN
    return CY_ERR_UNSUPPORTED;
}
