// Implicit-GEMM convolution (forward and dgrad) on gfx950 MFMA.
//
// GEMM view:  D[co][pixel] = sum_k W[co][k] * X[pixel][k],  k = (kh,kw,c).
// The weight tile is the MFMA A operand and the gathered activation tile the B operand, so a lane
// ends up with 4 consecutive output channels of one pixel -> one packed NHWC store.
//
// Block = 256 threads = 4 waves (2 over channels x 2 over pixels); tile BN channels x BM pixels;
// K step = one 128-byte LDS row per tile row (64 f16 / 32 f32).  Tiles go global -> LDS by
// buffer_load_dwordx4 ... lds in 1 KB pieces (an out-of-range offset = zero fill: padding, K tail,
// dgrad stride holes), double buffered, one barrier per K step.  LDS rows are 128 B with the 16-byte
// chunk index XOR-swizzled by (row & 7) -- applied on the source side, the DMA image being
// lane-linear -- conflict-free for the ds_read_b128 lane groups (see DESIGN.md section kernels).
// f16: v_mfma_f32_16x16x32_f16.  f32 (parity mode): v_mfma_f32_16x16x4_f32, exact f32.
//
// Two kernels share the scheme: igemm_fast_kernel (Cin a multiple of the K step: wave-uniform tap per
// step, SGPR offsets) and igemm_kernel (per-lane tap, any Cin: the first layers).  The 16-bit modes hand
// every launch that qualifies to the 8-wave pipelined kernel of conv_pipe.hip first; what stays here is
// the f32 parity mode, the fp32-output head convs, the first layers and grids too small for 256-pixel tiles.
#include <stdio.h>
#include <stdlib.h>

#include "igemm_common.hpp"

namespace {
using namespace cyk;

template <typename T, int BM, int BN>
__global__ void __launch_bounds__(256) igemm_kernel(const IgemmParams p) {
    constexpr int NW = 4;
    constexpr int CH = Elem<T>::CH;
    constexpr int BK = 8 * CH;
    constexpr int NT = NW * 64, RS = NW * 8;  // threads; tile rows staged per pass (8 rows of 128 B per wave instruction)
    constexpr int WMW = NW / 2;               // waves over pixels (2 waves over channels)
    static_assert(BM % RS == 0 && BN % RS == 0 && BM % (16 * WMW) == 0, "tile / wave layout");
    constexpr int XR = BM / RS, WR = BN / RS;
    constexpr int TI = BN / 32;          // channel fragments per wave (2 waves over BN)
    constexpr int TJ = BM / (16 * WMW);  // pixel fragments per wave   (WMW waves over BM)
    constexpr int STAGE = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 1, wm = wave >> 1;

    // XCD-aware tile order: blocks that share an X tile (same pixel tile, different channel tiles)
    // are made neighbours on one XCD's L2 (hardware places block b on XCD b % 8).
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, slot = bid >> 3;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    TapSet cls;
    const int tn = lid % p.ntiles, tm = select_class(p, lid / p.ntiles, cls);

    // ---- per-thread staging coordinates --------------------------------------------------------
    // direct-to-LDS: the LDS image of a wave-instruction is lane-linear (base + lane*16), so the swizzle moves to
    // the SOURCE: lane l fills physical chunk l&7 of row l>>3, i.e. it fetches logical chunk (l&7)^(row&7).
    const int chunk = (lane & 7) ^ ((lane >> 3) & 7);
    const int rbase = wave * 8 + (lane >> 3);
    // Per row (pixel) of this thread: byte offset of the "tap (0,0)" source pixel and a bit mask of the taps that fall
    // inside the image (and, for dgrad, on the stride lattice).  Per K step only  base + delta(tap) + channel  is left.
    const int ksign = p.transposed ? -1 : 1;
    const int sh = (p.transposed && p.stride == 2) ? 1 : 0;
    unsigned x_base[XR], x_mask[XR];
    const int ohw = p.OHc * p.OWc;
    {
        // one thread per pixel row works out (base, mask) once, the staging threads pick their XR rows up from LDS:
        // the two integer divisions and the tap loop run once per row instead of once per (row, staging thread)
        uint2* rowinfo = reinterpret_cast<uint2*>(smem);
        if (tid < BM) {
            const int m = tm * BM + tid;
            unsigned base = 0u, mask = 0u;
            if (m < p.M) {
                const int n = m / ohw, rem = m - n * ohw;
                const int ohc = rem / p.OWc;
                const int oh = ohc * p.oh_mul + cls.oh_off, ow = (rem - ohc * p.OWc) * p.ow_mul + cls.ow_off;
                const int xh = p.transposed ? oh + p.pad : oh * p.stride - p.pad;
                const int xw = p.transposed ? ow + p.pad : ow * p.stride - p.pad;
                for (int t = 0; t < cls.ntaps; ++t) {
                    const int kh = (cls.kh_pack >> (2 * t)) & 3, kw = (cls.kw_pack >> (2 * t)) & 3;
                    const int th = xh + ksign * kh, tw = xw + ksign * kw;
                    const bool ok = (((th | tw) & sh) == 0) & ((unsigned)(th >> sh) < (unsigned)p.GH) &
                                    ((unsigned)(tw >> sh) < (unsigned)p.GW);
                    mask |= (ok ? 1u : 0u) << t;
                }
                // for valid dgrad taps (th even) (xh - kh) >> 1 == (xh >> 1) - (kh >> 1), so the offset stays linear in the tap
                base = (unsigned)(((n * p.GH + (xh >> sh)) * p.GW + (xw >> sh)) * p.ldg) * (unsigned)sizeof(T);
            }
            rowinfo[tid] = make_uint2(base, mask);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const uint2 ri = rowinfo[rbase + RS * i];
            x_base[i] = ri.x;
            x_mask[i] = ri.y;
        }
        __syncthreads();   // stage 0 of the ring overlays rowinfo
    }
    const unsigned w_row0 = (unsigned)(tn * BN + rbase) * (unsigned)p.K * (unsigned)sizeof(T);
    const unsigned w_rstep = (unsigned)RS * (unsigned)p.K * (unsigned)sizeof(T);
    int k_c = chunk * CH, k_tap = 0;  // this thread's chunk: channel offset and tap within the K tile
    while (k_c >= p.GC) { k_c -= p.GC; ++k_tap; }
    const int ntaps = cls.ntaps;
    const int adv_tap = BK / p.GC, adv_c = BK - adv_tap * p.GC;
    // buffer descriptors: 32-bit byte offsets, and an out-of-range offset makes the DMA write zeros -- which is exactly
    // the zero fill padding / the K tail / dgrad holes need (a direct-to-LDS load cannot write a literal)
    const auto rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)p.g, 0, p.g_bytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFF00u;

    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto load_tile = [&](int kt, int stage) {
        unsigned char* xs_w = smem + stage * STAGE + wave_u * (8 * 128);
        unsigned char* ws_w = xs_w + BM * 128;
        const bool kvalid = k_tap < ntaps;
        const int tsh = 2 * min(k_tap, 15);
        const int kh = (cls.kh_pack >> tsh) & 3, kw = (cls.kw_pack >> tsh) & 3;
        // source offset of this tap relative to tap (0,0), launch-uniform geometry, per-thread tap only when GC < BK
        const unsigned tap_delta = (unsigned)(ksign * (((kh >> sh) * p.GW + (kw >> sh)) * p.ldg) + k_c) * (unsigned)sizeof(T);
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const bool ok = kvalid & ((x_mask[i] >> k_tap) & 1u);
            const unsigned off = ok ? x_base[i] + tap_delta : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (__attribute__((address_space(3))) void*)(xs_w + i * (RS * 128)),
                                                     16, off, 0, 0, 0);
        }
        const unsigned koff = (unsigned)((kh * p.ks + kw) * p.GC + k_c) * (unsigned)sizeof(T);  // column of the packed weights
        (void)kt;
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            const bool ok = kvalid & (tn * BN + rbase + RS * i < p.wrows);
            const unsigned off32 = ok ? w_row0 + i * w_rstep + koff : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(ws_w + i * (RS * 128)),
                                                     16, off32, 0, 0, 0);
        }
        // advance this thread's chunk to the next K tile (branch-free: BK = adv_tap * GC + adv_c)
        k_tap += adv_tap;
        k_c += adv_c;
        const bool wrap = k_c >= p.GC;
        k_c -= wrap ? p.GC : 0;
        k_tap += wrap ? 1 : 0;
    };
    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkt = (cls.ntaps * p.GC + BK - 1) / BK;
    auto compute_tile = [&](int stage) {
        const unsigned char* xs = smem + stage * STAGE;
        const unsigned char* ws = xs + BM * 128;
        const unsigned char* wrow = ws + (wn * (BN / 2) + (lane & 15)) * 128;
        const unsigned char* xrow = xs + (wm * (BM / WMW) + (lane & 15)) * 128;
#pragma unroll
        for (int kk = 0; kk < Mma<T>::KSTEPS; ++kk) {
            typename Mma<T>::frag a[TI], b[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) a[i] = Mma<T>::load(wrow + i * 16 * 128, kk, lane);
#pragma unroll
            for (int j = 0; j < TJ; ++j) b[j] = Mma<T>::load(xrow + j * 16 * 128, kk, lane);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = Mma<T>::mma(a[i], b[j], acc[i][j]);
        }
    };
    load_tile(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) load_tile(kt + 1, cur ^ 1);
        compute_tile(cur);
        __syncthreads();
    }

    igemm_epilogue<T, BM, BN, NW, false>(p, cls, acc, tm, tn, lid, smem);
}

// ---------------------------------------------------------------------------------------------------------------------
// Fast path of the same tile scheme for GC % BK == 0 (every layer but the first ones): all lanes of a K step share the
// tap and the channel offset, so the per-step part of every source address is wave-uniform and rides in the buffer
// instruction's SGPR offset.  What is left per DMA piece is one sign-extending bit-field extract and one OR on a
// per-thread constant (an invalid (row, tap) turns the VGPR offset into 0xFFFFFFFF = out of range = zero fill): 2 VALU
// per activation piece and none per weight piece, against ~7 per piece in the general kernel above (measured: 1-5 %;
// spreading the pieces between the MFMA groups instead of issuing them in a burst measured the same within noise).
// Range checking only sees the VGPR offset, so the descriptor base sits x_bias bytes BELOW the tensor: the VGPR part
// x_base + x_bias + dmin is then non-negative for every row that has a valid tap, and the SGPR part delta(tap) - dmin >= 0.
template <typename T, int BM, int BN>
__global__ void __launch_bounds__(256, 2) igemm_fast_kernel(const IgemmParams p) {
    constexpr int CH = Elem<T>::CH;
    constexpr int BK = 8 * CH;
    constexpr int NW = 4, RS = NW * 8, WMW = NW / 2;
    constexpr int XR = BM / RS, WR = BN / RS, NP = XR + WR;
    constexpr int TI = BN / 32, TJ = BM / (16 * WMW);
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int KS = Mma<T>::KSTEPS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 1, wm = wave >> 1;
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, slot = bid >> 3;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    TapSet cls;
    const int tn = lid % p.ntiles, tm = select_class(p, lid / p.ntiles, cls);

    const int chunk = (lane & 7) ^ ((lane >> 3) & 7);
    const int rbase = wave * 8 + (lane >> 3);
    const int ksign = p.transposed ? -1 : 1;
    const int sh = (p.transposed && p.stride == 2) ? 1 : 0;
    const int span = (p.ks - 1) >> sh;
    const int dmin = p.transposed ? -(span * p.GW + span) * p.ldg * (int)sizeof(T) : 0;
    unsigned xoff[XR];
    int ximask[XR];
    const int ohw = p.OHc * p.OWc;
    if (p.ks == 1 && p.stride == 1 && p.pad == 0 && ohw == p.OH * p.OW && (p.oh_mul | p.ow_mul) == 1) {
        // 1x1 / stride 1: the source pixel IS the output pixel -- no divisions, no LDS hand-over, no barriers
        const unsigned lane_const = (unsigned)p.x_bias + (unsigned)(chunk * CH) * (unsigned)sizeof(T);
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int m = tm * BM + rbase + RS * i;
            xoff[i] = (unsigned)m * (unsigned)p.ldg * (unsigned)sizeof(T) + lane_const;
            ximask[i] = m < p.M ? ~1 : ~0;
        }
    } else {
        uint2* rowinfo = reinterpret_cast<uint2*>(smem);
        if (tid < BM) {
            const int m = tm * BM + tid;
            unsigned base = 0u, mask = 0u;
            if (m < p.M) {
                const int n = m / ohw, rem = m - n * ohw;
                const int ohc = rem / p.OWc;
                const int oh = ohc * p.oh_mul + cls.oh_off, ow = (rem - ohc * p.OWc) * p.ow_mul + cls.ow_off;
                const int xh = p.transposed ? oh + p.pad : oh * p.stride - p.pad;
                const int xw = p.transposed ? ow + p.pad : ow * p.stride - p.pad;
                for (int t = 0; t < cls.ntaps; ++t) {
                    const int kh = (cls.kh_pack >> (2 * t)) & 3, kw = (cls.kw_pack >> (2 * t)) & 3;
                    const int th = xh + ksign * kh, tw = xw + ksign * kw;
                    const bool ok = (((th | tw) & sh) == 0) & ((unsigned)(th >> sh) < (unsigned)p.GH) &
                                    ((unsigned)(tw >> sh) < (unsigned)p.GW);
                    mask |= (ok ? 1u : 0u) << t;
                }
                base = (unsigned)(((n * p.GH + (xh >> sh)) * p.GW + (xw >> sh)) * p.ldg) * (unsigned)sizeof(T);
            }
            rowinfo[tid] = make_uint2(base, mask);
        }
        __syncthreads();
        const unsigned lane_const = (unsigned)p.x_bias + (unsigned)dmin + (unsigned)(chunk * CH) * (unsigned)sizeof(T);
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const uint2 ri = rowinfo[rbase + RS * i];
            xoff[i] = ri.x + lane_const;
            ximask[i] = (int)~ri.y;
        }
        __syncthreads();
    }
    unsigned woff[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int row = tn * BN + rbase + RS * i;
        woff[i] = row < p.wrows ? ((unsigned)row * (unsigned)p.K + (unsigned)(chunk * CH)) * (unsigned)sizeof(T) : 0xFFFFFFFFu;
    }
    const auto rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)(p.g - p.x_bias), 0, p.g_bytes + p.x_bias, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    // wave-uniform K position: tap index and channel offset of the tile being LOADED
    int l_tap = 0, l_c = 0;
    unsigned x_soff = 0u, w_soff = 0u;
    auto next_offsets = [&]() {
        const int tsh = 2 * l_tap;
        const int kh = (cls.kh_pack >> tsh) & 3, kw = (cls.kw_pack >> tsh) & 3;
        x_soff = (unsigned)((ksign * (((kh >> sh) * p.GW + (kw >> sh)) * p.ldg) + l_c) * (int)sizeof(T) - dmin);
        w_soff = (unsigned)(((kh * p.ks + kw) * p.GC + l_c) * (int)sizeof(T));
    };
    auto advance = [&]() {
        l_c += BK;
        if (l_c >= p.GC) { l_c = 0; ++l_tap; }
    };
    auto issue_piece = [&](int pc, int stage, int tap) {
        unsigned char* xs_w = smem + stage * STAGE + wave_u * (8 * 128);
        if (pc < XR) {
            const unsigned v = xoff[pc] | (unsigned)(-((ximask[pc] >> tap) & 1));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (__attribute__((address_space(3))) void*)(xs_w + pc * (RS * 128)), 16,
                                                     v, x_soff, 0, 0);
        } else {
            const unsigned v = woff[pc - XR];   // (a captured array element passed straight to the builtin makes hipcc's
                                                // host pass drop the whole kernel stub without a diagnostic)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rs_w, (__attribute__((address_space(3))) void*)(xs_w + BM * 128 + (pc - XR) * (RS * 128)), 16, v, w_soff, 0, 0);
        }
    };

    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkt = cls.ntaps * (p.GC / BK);
    next_offsets();
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) issue_piece(pc, 0, 0);
    advance();
    __syncthreads();

    auto step = [&](int cur, const bool LOAD) {
        const unsigned char* xs = smem + cur * STAGE;
        const unsigned char* ws = xs + BM * 128;
        const unsigned char* wrow = ws + (wn * (BN / 2) + (lane & 15)) * 128;
        const unsigned char* xrow = xs + (wm * (BM / WMW) + (lane & 15)) * 128;
        const int tap_next = l_tap;   // tap of the tile being loaded (the state runs one tile ahead of the MFMAs)
        if (LOAD) next_offsets();
        if (LOAD) {
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) issue_piece(pc, cur ^ 1, tap_next);
        }
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            typename Mma<T>::frag a[TI], b[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) a[i] = Mma<T>::load(wrow + i * 16 * 128, kk, lane);
#pragma unroll
            for (int j = 0; j < TJ; ++j) b[j] = Mma<T>::load(xrow + j * 16 * 128, kk, lane);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = Mma<T>::mma(a[i], b[j], acc[i][j]);
        }
        if (LOAD) advance();
    };
    for (int kt = 0; kt + 1 < nkt; ++kt) {
        step(kt & 1, true);
        __syncthreads();
    }
    step((nkt - 1) & 1, false);
    __syncthreads();

    igemm_epilogue<T, BM, BN, NW, false>(p, cls, acc, tm, tn, lid, smem);
}

template <typename T, int BM, int BN>
int launch_general(const IgemmParams& p0, hipStream_t s) {
    IgemmParams p = p0;
    p.mtiles = ((p.M + BM - 1) / BM) * p.ncls;
    p.ntiles = (p.OC + BN - 1) / BN;
    constexpr int smem = 2 * (BM + BN) * 128;
    static unsigned long long attr_done = 0;      // bit d: set for HIP device d
    if (cy_first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<T, BM, BN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
    hipLaunchKernelGGL((igemm_kernel<T, BM, BN>), dim3(p.mtiles * p.ntiles), dim3(256), smem, s, p);
    CY_LAUNCH_CHECK();
    return 0;
}

template <typename T, int BM, int BN>
int launch_fast(const IgemmParams& p0, hipStream_t s) {
    IgemmParams p = p0;
    p.mtiles = ((p.M + BM - 1) / BM) * p.ncls;
    p.ntiles = (p.OC + BN - 1) / BN;
    constexpr int smem = 2 * (BM + BN) * 128;
    static unsigned long long attr_done = 0;      // bit d: set for HIP device d
    if (cy_first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_fast_kernel<T, BM, BN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
    hipLaunchKernelGGL((igemm_fast_kernel<T, BM, BN>), dim3(p.mtiles * p.ntiles), dim3(256), smem, s, p);
    CY_LAUNCH_CHECK();
    return 0;
}

template <typename T, int BM, int BN>
int launch(const IgemmParams& p, hipStream_t s) {
    constexpr int BK = 8 * Elem<T>::CH;
    if (p.x_bias && p.GC % BK == 0) return launch_fast<T, BM, BN>(p, s);   // wave-uniform tap per K step
    return launch_general<T, BM, BN>(p, s);
}

// Tile choice.  Channel tile = min(128, OC rounded up to 32).  Pixel tile: blocks run in rounds of 256 * blocks-per-CU
// (LDS-limited) and a K step costs ~1 us per resident block almost independently of the tile area (the double-buffered
// loop is latency-bound; CY_IGEMM_TILE sweeps with tools/conv_micro.py), so
//   * a problem that does not even fill one round of 128-pixel tiles uses 64-pixel tiles (more blocks in flight);
//   * a problem that needs several rounds uses 192-pixel tiles when that saves a round (e.g. the 76x76 layers at
//     batch 16: 722 tiles = 2 rounds at 128, 482 tiles = 1 round at 192; measured in-model 0.43 -> 0.375 ms).
inline long tile_rounds(int M, int OC, int bm, int bn) {
    int per_cu = (160 * 1024) / (2 * (bm + bn) * 128);
    if (per_cu > 8) per_cu = 8;
    const long tiles = (long)((M + bm - 1) / bm) * ((OC + bn - 1) / bn);
    return (tiles + 256L * per_cu - 1) / (256L * per_cu);
}
inline void pick_tile(int M, int OC, int K, int esize, int& bm, int& bn) {
    (void)K; (void)esize;
    bn = OC > 64 ? 128 : (OC > 32 ? 64 : 32);
    bm = 128;
    const long blocks128 = (long)((M + 127) / 128) * ((OC + bn - 1) / bn);
    if (blocks128 < 512) {
        bm = 64;
        // still under one round of 64 x 128 tiles (the 19 x 19 layers): halve the channel tile too -- 512->512 @19x19
        // 50.9 -> 46.2 us, the dgrad of 512->1024 94.6 -> 89.3 us
        if (bn == 128 && (long)((M + 63) / 64) * ((OC + 127) / 128) < 512) bn = 64;
    } else if (bn >= 64 && tile_rounds(M, OC, 192, bn) < tile_rounds(M, OC, 128, bn)) {
        bm = 192;
    }
}

template <typename T>
int dispatch_tiles(const IgemmParams& p, hipStream_t s) {
    int bm, bn;
    pick_tile(p.M * p.ncls, p.OC, p.ntaps * p.GC, (int)sizeof(T), bm, bn);
#define CY_TILE(BM_, BN_) \
    if (bm == BM_ && bn == BN_) return launch<T, BM_, BN_>(p, s);
    CY_TILE(128, 128) CY_TILE(128, 64) CY_TILE(128, 32) CY_TILE(64, 128) CY_TILE(64, 64) CY_TILE(64, 32)
    CY_TILE(192, 128) CY_TILE(192, 64)
#undef CY_TILE
    return CY_ERR_ARG;
}

// one launch: the pipelined kernel when the shape qualifies, else the 4-wave kernels of this file
int dispatch(const IgemmParams& p, int dtype, hipStream_t s) {
    int used = 0;
    int rc = (p.flags & CY_CONV_BN_FUSED) ? 0 : cy_direct_try(p, dtype, s, &used);
    if (rc || used) return rc;
    rc = cy_pipe_try(p, dtype, s, &used);
    if (rc || used) return rc;
    if (p.flags & CY_CONV_BN_FUSED) return CY_ERR_UNSUPPORTED;   // two-phase epilogue: the pipelined kernel, one round, or not at all
    if (p.flags & CY_CONV_BNBWD_SUMS) return CY_ERR_ARG;   // only the pipelined kernel has that epilogue
    if (dtype == CY_F16) return dispatch_tiles<f16>(p, s);
    if (dtype == CY_BF16) return dispatch_tiles<bf16>(p, s);
    return dispatch_tiles<float>(p, s);
}

}  // namespace

extern "C" int cy_conv_stats_rows(int M, int OC) {
    (void)M; (void)OC;
    return CY_STAT_BINS;
}

// CY_CONV_STATS_DET: one table row per pixel tile; no kernel uses tiles of fewer than 64 pixels
extern "C" int cy_conv_stats_rows_det(int M, int OC) {
    (void)OC;
    return (M + 63) / 64;
}

static int conv_igemm_impl(const void* g, int N, int GH, int GW, int GC, int ldg, const void* w, int wrows, void* out,
                           int OH, int OW, int OC, int ldo, int ks, int stride, int pad, int dtype, int flags,
                           const float* bias, float* stats_part, int* stats_rows_host, const float* aff_scale,
                           const float* aff_shift, int act, const void* res, int ldres, cy_stream_t s,
                           const float* bn_mean = nullptr, const float* bn_invstd = nullptr, const IgemmParams* fuse = nullptr) {
    const int ch = dtype == CY_F32 ? 4 : 8;
    if (!g || !w || !out || (dtype != CY_F16 && dtype != CY_BF16 && dtype != CY_F32)) return CY_ERR_ARG;
    if ((ks != 1 && ks != 3) || (stride != 1 && stride != 2) || GC % ch || ldg % ch) return CY_ERR_ARG;
    if ((flags & CY_CONV_STATS) && !stats_part) return CY_ERR_ARG;
    if ((flags & CY_CONV_STATS_DET) && !(flags & (CY_CONV_STATS | CY_CONV_BNBWD_SUMS))) return CY_ERR_ARG;
    if (!(flags & CY_CONV_BIAS_F32OUT) && (ldo % 4)) return CY_ERR_ARG;
    IgemmParams p;
    p.o2 = nullptr; p.ldo2 = 0; p.bn_gamma = p.bn_beta = nullptr; p.bn_rmean = p.bn_rvar = nullptr; p.bn_nbt = nullptr;
    p.bn_momentum = p.bn_eps = 0.f; p.bn_vec = p.bn_zero = nullptr; p.bn_zero_n = 0; p.ticket = nullptr;
    if (fuse) {
        p.o2 = fuse->o2; p.ldo2 = fuse->ldo2; p.bn_gamma = fuse->bn_gamma; p.bn_beta = fuse->bn_beta; p.bn_rmean = fuse->bn_rmean;
        p.bn_rvar = fuse->bn_rvar; p.bn_nbt = fuse->bn_nbt; p.bn_momentum = fuse->bn_momentum; p.bn_eps = fuse->bn_eps;
        p.bn_vec = fuse->bn_vec; p.bn_zero = fuse->bn_zero; p.bn_zero_n = fuse->bn_zero_n; p.ticket = fuse->ticket;
    }
    p.g = (const unsigned char*)g; p.w = (const unsigned char*)w; p.o = (unsigned char*)out;
    p.bias = bias; p.stats = stats_part;
    p.aff_scale = aff_scale; p.aff_shift = aff_shift; p.act = act; p.res = (const unsigned char*)res; p.ldres = ldres;
    p.bn_mean = bn_mean; p.bn_invstd = bn_invstd;
    if ((flags & CY_CONV_AFFINE_ACT) && (!aff_scale || !aff_shift || (flags & (CY_CONV_STATS | CY_CONV_TRANSPOSED)))) return CY_ERR_ARG;
    p.N = N; p.GH = GH; p.GW = GW; p.GC = GC; p.ldg = ldg;
    p.OH = OH; p.OW = OW; p.OC = OC; p.ldo = ldo;
    p.ks = ks; p.stride = stride; p.pad = pad; p.transposed = (flags & CY_CONV_TRANSPOSED) ? 1 : 0;
    p.K = ks * ks * GC; p.M = N * OH * OW; p.wrows = wrows; p.flags = flags;
    p.mtiles = p.ntiles = 0; p.bm_eff = 0; p.slab_rows = 0;
    p.stat_det = (flags & CY_CONV_STATS_DET) ? 1 : 0;
    if (p.M <= 0 || OC <= 0) return CY_ERR_ARG;
    if ((flags & CY_CONV_BNBWD_SUMS) && (!bn_mean || (flags & (CY_CONV_STATS | CY_CONV_AFFINE_ACT | CY_CONV_BIAS_F32OUT))))
        return CY_ERR_ARG;      // (stride 2: only the direct small-channel kernel has the epilogue; dispatch() says so)
    if (stats_rows_host) *stats_rows_host = (flags & CY_CONV_STATS_DET) ? cy_conv_stats_rows_det(p.M, OC) : cy_conv_stats_rows(p.M, OC);
    const size_t esz = dtype == CY_F32 ? 4 : 2;
    const size_t gb = (((size_t)N * GH * GW - 1) * ldg + GC) * esz, wb = (size_t)wrows * p.K * esz;
    if (gb >= 0xFFFFFF00ull || wb >= 0xFFFFFF00ull) return CY_ERR_ARG;  // 32-bit buffer offsets
    p.g_bytes = (unsigned)gb; p.w_bytes = (unsigned)wb;
    {
        const size_t bias = (size_t)(2 * GW + 2) * ldg * esz;   // see igemm_fast_kernel
        p.x_bias = gb + bias < 0xFFFFFF00ull ? (unsigned)bias : 0u;
    }
    p.OHc = OH; p.OWc = OW; p.oh_mul = p.ow_mul = 1; p.oh_off = p.ow_off = 0;
    p.ncls = 1;
    p.ntaps = ks * ks; p.kh_pack = p.kw_pack = 0;
    for (int t = 0; t < ks * ks; ++t) {
        p.kh_pack |= (unsigned)(t / ks) << (2 * t);
        p.kw_pack |= (unsigned)(t % ks) << (2 * t);
    }
    if (!(p.transposed && stride == 2)) return dispatch(p, dtype, cy_s(s));
    // stride-2 dgrad: an input-gradient pixel only sees the taps with (o + pad - k) even: four (row, column) parity classes,
    // each over its own taps -- 9 tap-visits in total instead of 36.
    IgemmParams cls[4];
    int ncls = 0;
    for (int ph = 1; ph >= 0; --ph)          // (1,1), (1,0), (0,1), (0,0): 4, 2, 2, 1 taps for a 3x3 / pad 1 kernel
        for (int pw = 1; pw >= 0; --pw) {
            IgemmParams q = p;
            q.OHc = (OH - ph + 1) / 2; q.OWc = (OW - pw + 1) / 2;
            if (q.OHc <= 0 || q.OWc <= 0) continue;
            q.oh_mul = q.ow_mul = 2; q.oh_off = ph; q.ow_off = pw;
            q.M = N * q.OHc * q.OWc;
            q.ntaps = 0; q.kh_pack = q.kw_pack = 0;
            for (int kh = 0; kh < ks; ++kh) {
                if ((ph + pad - kh) & 1) continue;
                for (int kw = 0; kw < ks; ++kw) {
                    if ((pw + pad - kw) & 1) continue;
                    q.kh_pack |= (unsigned)kh << (2 * q.ntaps);
                    q.kw_pack |= (unsigned)kw << (2 * q.ntaps);
                    ++q.ntaps;
                }
            }
            if (q.ntaps == 0) {
                // (1x1 stride 2: only the (pad, pad) class receives anything; the others are zeros the caller's store / accumulate
                // semantics still expect to be written) -- keep the class, a tap-less launch stores zeros
            }
            cls[ncls++] = q;
        }
    // ONE launch when the four sub-lattices are congruent (OH, OW even -- every stride-2 conv of the Darknet cfgs) and every
    // class has a tap: the grid interleaves the classes tile by tile, so a heavy (4-tap) and a light (1-tap) tile sit next to
    // each other on every XCD and the launch pays one ramp and one tail instead of four.  CY_DGRAD_S2_MERGE=0: four launches.
    static int merge = -1;
    if (merge < 0) { const char* e = getenv("CY_DGRAD_S2_MERGE"); merge = e ? atoi(e) : 1; }
    bool can = merge && ncls == 4 && !(OH & 1) && !(OW & 1);
    for (int c = 0; c < ncls && can; ++c) can = cls[c].ntaps > 0;
    if (can) {
        IgemmParams q = cls[0];      // (the kernels derive every class's offsets and taps from its number: select_class)
        q.ncls = 4;
        return dispatch(q, dtype, cy_s(s));
    }
    if (flags & CY_CONV_BNBWD_SUMS) return CY_ERR_ARG;      // the sums ride on a single launch only
    for (int c = 0; c < ncls; ++c) {
        const int rc = dispatch(cls[c], dtype, cy_s(s));
        if (rc) return rc;
    }
    return 0;
}

extern "C" int cy_conv_igemm(const void* g, int N, int GH, int GW, int GC, int ldg, const void* w, int wrows,
                             void* out, int OH, int OW, int OC, int ldo, int ks, int stride, int pad, int dtype,
                             int flags, const float* bias, float* stats_part, int* stats_rows_host, cy_stream_t s) {
    CY_ENTER();
    if (flags & (CY_CONV_AFFINE_ACT | CY_CONV_BNBWD_SUMS)) return CY_ERR_ARG;
    return conv_igemm_impl(g, N, GH, GW, GC, ldg, w, wrows, out, OH, OW, OC, ldo, ks, stride, pad, dtype, flags, bias,
                           stats_part, stats_rows_host, nullptr, nullptr, 0, nullptr, 0, s);
}

extern "C" int cy_conv_bn_act_eval(const void* g, int N, int GH, int GW, int GC, int ldg, const void* w, int wrows,
                                   void* out, int OH, int OW, int OC, int ldo, int ks, int stride, int pad, int dtype,
                                   const float* scale, const float* shift, int act, const void* res, int ldres,
                                   int flags, cy_stream_t s) {
    CY_ENTER();
    if (flags & ~CY_CONV_TILE(15)) return CY_ERR_ARG;      // only the kernel / tile hint is accepted here
    return conv_igemm_impl(g, N, GH, GW, GC, ldg, w, wrows, out, OH, OW, OC, ldo, ks, stride, pad, dtype,
                           CY_CONV_AFFINE_ACT | flags, nullptr, nullptr, nullptr, scale, shift, act, res, ldres, s);
}

extern "C" int cy_conv_bn_act_train(const void* g, int N, int GH, int GW, int GC, int ldg, const void* w, int wrows, void* raw,
                                    int OH, int OW, int OC, int ldraw, void* out, int ldout, const void* res, int ldres, int ks,
                                    int stride, int pad, int dtype, int flags, float* stats_bins, const float* gamma,
                                    const float* beta, float* running_mean, float* running_var, void* num_batches_tracked,
                                    float momentum, float eps, float* vec, float* zero_table, int zero_n, int act,
                                    int32_t* ticket, cy_stream_t s) {
    CY_ENTER();
    if (flags & ~CY_CONV_TILE(15)) return CY_ERR_ARG;      // only the kernel / tile hint is accepted here
    if (!out || !stats_bins || !gamma || !beta || !vec || !ticket || (zero_n > 0 && !zero_table) || zero_table == stats_bins ||
        (running_mean && !running_var) || ldout % 8 || (res && ldres % 8))
        return CY_ERR_ARG;
    if (dtype != CY_F16 && dtype != CY_BF16) return CY_ERR_UNSUPPORTED;
    IgemmParams f;
    f.o2 = (unsigned char*)out; f.ldo2 = ldout; f.bn_gamma = gamma; f.bn_beta = beta; f.bn_rmean = running_mean;
    f.bn_rvar = running_var; f.bn_nbt = (long long*)num_batches_tracked; f.bn_momentum = momentum; f.bn_eps = eps; f.bn_vec = vec;
    f.bn_zero = zero_table; f.bn_zero_n = zero_n; f.ticket = ticket;
    return conv_igemm_impl(g, N, GH, GW, GC, ldg, w, wrows, raw, OH, OW, OC, ldraw, ks, stride, pad, dtype,
                           CY_CONV_STATS | CY_CONV_BN_FUSED | flags, nullptr, stats_bins, nullptr, nullptr, nullptr, act, res, ldres, s,
                           nullptr, nullptr, &f);
}

extern "C" int cy_conv_dgrad_bn_sums(const void* g, int N, int GH, int GW, int GC, int ldg, const void* w, int wrows,
                                     void* out, int OH, int OW, int OC, int ldo, int ks, int stride, int pad, int dtype,
                                     int flags, const void* raw, int ldraw, const float* mean, const float* invstd,
                                     const float* scale, const float* shift, int act, float* sums_part, cy_stream_t s) {
    CY_ENTER();
    if (flags & ~(CY_CONV_TRANSPOSED | CY_CONV_ACCUM | CY_CONV_STATS_DET | CY_CONV_TILE(15))) return CY_ERR_ARG;
    if (!raw || !mean || !invstd || !scale || !shift || !sums_part || ldraw % 8 || ((uintptr_t)raw & 15)) return CY_ERR_ARG;
    if (dtype != CY_F16 && dtype != CY_BF16) return CY_ERR_ARG;
    return conv_igemm_impl(g, N, GH, GW, GC, ldg, w, wrows, out, OH, OW, OC, ldo, ks, stride, pad, dtype,
                           flags | CY_CONV_BNBWD_SUMS, nullptr, sums_part, nullptr, scale, shift, act, raw, ldraw, s, mean,
                           invstd);
}
