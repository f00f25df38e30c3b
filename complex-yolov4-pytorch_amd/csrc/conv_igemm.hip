// Implicit-GEMM convolution (forward and dgrad) on gfx950 MFMA.
//
// GEMM view:  D[co][pixel] = sum_k W[co][k] * X[pixel][k],  k = (kh,kw,c).
// The weight tile is the MFMA A operand and the gathered activation tile the B operand, so a lane
// ends up with 4 consecutive output channels of one pixel -> one packed NHWC store.
//
// Block = 256 threads = 4 waves (2 over channels x 2 over pixels); tile BN channels x BM pixels;
// K step = one 128-byte LDS row per tile row (64 f16 / 32 f32).  Tiles are staged
// global -> VGPR -> LDS (zero fill for padding / K tail / dgrad stride holes), double buffered, one
// barrier per K step.  LDS rows are 128 B with the 16-byte chunk index XOR-swizzled by (row & 7):
// conflict-free for the ds_read_b128 lane groups (see DESIGN.md section kernels).
// f16: v_mfma_f32_16x16x32_f16.  f32 (parity mode): v_mfma_f32_16x16x4_f32, exact f32.
#include <stdio.h>
#include <stdlib.h>

#include "common.hpp"

namespace {

struct IgemmParams {
    const unsigned char* g;
    const unsigned char* w;
    unsigned char* o;
    const float* bias;
    float* stats;
    int N, GH, GW, GC, ldg;
    int OH, OW, OC, ldo;
    int ks, stride, pad, transposed;
    int K, M, wrows;
    int flags;
    int mtiles, ntiles;
    // pixel sub-lattice handled by this launch: oh = oh' * oh_mul + oh_off over OHc x OWc (the whole image for
    // ordinary launches; one parity class for a stride-2 dgrad) and its taps, 2 bits per (kh, kw)
    int OHc, OWc, oh_mul, oh_off, ow_mul, ow_off;
    int ntaps;
    unsigned kh_pack, kw_pack;
    unsigned g_bytes, w_bytes;  // extents of the gathered view / weight matrix (buffer descriptors: OOB reads return 0)
    // CY_CONV_AFFINE_ACT epilogue (eval mode): out = act(acc * aff_scale[co] + aff_shift[co]) (+ res)
    const float* aff_scale;
    const float* aff_shift;
    const unsigned char* res;
    int act, ldres;
    int dbg_nomma;   // CY_IGEMM_NOMMA=1: skip the MFMA phase (load-path ceiling experiment)
};

template <typename T>
struct Mma;
template <>
struct Mma<f16> {
    static constexpr int KSTEPS = 2;  // MFMA steps per 128-byte LDS row
    typedef f16x8 frag;
    __device__ static __forceinline__ frag load(const unsigned char* row_ptr, int kk, int lane) {
        const int c = (kk * 4 + (lane >> 4)) ^ (lane & 7);
        return *reinterpret_cast<const frag*>(row_ptr + (c << 4));
    }
    __device__ static __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
template <>
struct Mma<float> {
    static constexpr int KSTEPS = 8;
    typedef float frag;
    __device__ static __forceinline__ frag load(const unsigned char* row_ptr, int kk, int lane) {
        const int c = kk ^ (lane & 7);
        return *reinterpret_cast<const float*>(row_ptr + (c << 4) + ((lane >> 4) << 2));
    }
    __device__ static __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
};

// sum over the 16 lanes of a DPP row (every lane of the row ends up with the total)
__device__ __forceinline__ float row16_sum(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror
    return v;
}

template <typename T, int BM, int BN, bool GLDS, int NST = 2>
__global__ void __launch_bounds__(256) igemm_kernel(const IgemmParams p) {
    static_assert(NST == 2 || GLDS, "the 3-stage ring needs direct-to-LDS loads");
    constexpr int CH = Elem<T>::CH;
    constexpr int BK = 8 * CH;
    constexpr int XR = BM / 32, WR = BN / 32;
    constexpr int TI = BN / 32;  // channel fragments per wave (2 waves over BN)
    constexpr int TJ = BM / 32;  // pixel fragments per wave   (2 waves over BM)
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
    const int tn = lid % p.ntiles, tm = lid / p.ntiles;

    // ---- per-thread staging coordinates --------------------------------------------------------
    // register staging: thread = (row tid>>3, chunk tid&7), swizzle applied on the LDS store.
    // direct-to-LDS (GLDS): the LDS image of a wave-instruction is lane-linear (base + lane*16), so the swizzle moves to
    // the SOURCE: lane l fills physical chunk l&7 of row l>>3, i.e. it fetches logical chunk (l&7)^(row&7).
    const int chunk = GLDS ? ((lane & 7) ^ ((p.dbg_nomma & 2) ? 0 : ((lane >> 3) & 7))) : (tid & 7);   // dbg bit 1: no source swizzle (timing experiment only)
    const int rbase = GLDS ? (wave * 8 + (lane >> 3)) : (tid >> 3);
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
                const int oh = ohc * p.oh_mul + p.oh_off, ow = (rem - ohc * p.OWc) * p.ow_mul + p.ow_off;
                const int xh = p.transposed ? oh + p.pad : oh * p.stride - p.pad;
                const int xw = p.transposed ? ow + p.pad : ow * p.stride - p.pad;
                for (int t = 0; t < p.ntaps; ++t) {
                    const int kh = (p.kh_pack >> (2 * t)) & 3, kw = (p.kw_pack >> (2 * t)) & 3;
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
            const uint2 ri = rowinfo[rbase + 32 * i];
            x_base[i] = ri.x;
            x_mask[i] = ri.y;
        }
        __syncthreads();   // stage 0 of the ring overlays rowinfo
    }
    const unsigned w_row0 = (unsigned)(tn * BN + rbase) * (unsigned)p.K * (unsigned)sizeof(T);
    const unsigned w_rstep = 32u * (unsigned)p.K * (unsigned)sizeof(T);
    int k_c = chunk * CH, k_tap = 0;  // this thread's chunk: channel offset and tap within the K tile
    while (k_c >= p.GC) { k_c -= p.GC; ++k_tap; }
    const int ntaps = p.ntaps;
    const int adv_tap = BK / p.GC, adv_c = BK - adv_tap * p.GC;
    // buffer descriptors: 32-bit byte offsets, and an out-of-range offset makes the DMA write zeros -- which is exactly
    // the zero fill padding / the K tail / dgrad holes need (a direct-to-LDS load cannot write a literal)
    const auto rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)p.g, 0, p.g_bytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFF00u;
    (void)rs_g; (void)rs_w;

    u32x4 xv[XR], wv[WR];
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto load_tile = [&](int kt, int stage) {
        unsigned char* xs_w = smem + stage * STAGE + wave_u * (8 * 128);
        unsigned char* ws_w = xs_w + BM * 128;
        (void)xs_w; (void)ws_w;
        const bool kvalid = k_tap < ntaps;
        const int tsh = 2 * min(k_tap, 15);
        const int kh = (p.kh_pack >> tsh) & 3, kw = (p.kw_pack >> tsh) & 3;
        // source offset of this tap relative to tap (0,0), launch-uniform geometry, per-thread tap only when GC < BK
        const unsigned tap_delta = (unsigned)(ksign * (((kh >> sh) * p.GW + (kw >> sh)) * p.ldg) + k_c) * (unsigned)sizeof(T);
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const bool ok = kvalid & ((x_mask[i] >> k_tap) & 1u);
            if constexpr (GLDS) {
                const unsigned off = ok ? x_base[i] + tap_delta : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (__attribute__((address_space(3))) void*)(xs_w + i * (32 * 128)),
                                                         16, off, 0, 0, 0);
            } else {
                u32x4 v = {0u, 0u, 0u, 0u};
                if (ok) v = *reinterpret_cast<const u32x4*>(p.g + (size_t)(x_base[i] + tap_delta));
                xv[i] = v;
            }
        }
        const unsigned koff = (unsigned)((kh * p.ks + kw) * p.GC + k_c) * (unsigned)sizeof(T);  // column of the packed weights
        (void)kt;
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            const bool ok = kvalid & (tn * BN + rbase + 32 * i < p.wrows);
            if constexpr (GLDS) {
                const unsigned off32 = ok ? w_row0 + i * w_rstep + koff : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(ws_w + i * (32 * 128)),
                                                         16, off32, 0, 0, 0);
            } else {
                u32x4 v = {0u, 0u, 0u, 0u};
                if (ok) v = *reinterpret_cast<const u32x4*>(p.w + (size_t)(w_row0 + i * w_rstep + koff));
                wv[i] = v;
            }
        }
        // advance this thread's chunk to the next K tile (branch-free: BK = adv_tap * GC + adv_c)
        k_tap += adv_tap;
        k_c += adv_c;
        const bool wrap = k_c >= p.GC;
        k_c -= wrap ? p.GC : 0;
        k_tap += wrap ? 1 : 0;
    };
    auto store_tile = [&](int stage) {
        if constexpr (GLDS) return;
        unsigned char* xs = smem + stage * STAGE;
        unsigned char* ws = xs + BM * 128;
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int row = rbase + 32 * i;
            *reinterpret_cast<u32x4*>(xs + row * 128 + ((chunk ^ (row & 7)) << 4)) = xv[i];
        }
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            const int row = rbase + 32 * i;
            *reinterpret_cast<u32x4*>(ws + row * 128 + ((chunk ^ (row & 7)) << 4)) = wv[i];
        }
    };

    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkt = (p.dbg_nomma & 16) ? 1 : (p.ntaps * p.GC + BK - 1) / BK;   // dbg bit 4: one K step (fixed-cost experiment)
    auto compute_tile = [&](int stage) {
        const unsigned char* xs = smem + stage * STAGE;
        const unsigned char* ws = xs + BM * 128;
        const unsigned char* wrow = ws + (wn * (BN / 2) + (lane & 15)) * 128;
        const unsigned char* xrow = xs + (wm * (BM / 2) + (lane & 15)) * 128;
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
    if constexpr (NST == 2) {
        load_tile(0, 0);
        store_tile(0);
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nkt) load_tile(kt + 1, cur ^ 1);
            if (!(p.dbg_nomma & 1)) compute_tile(cur);
            if (kt + 1 < nkt) store_tile(cur ^ 1);
            __syncthreads();
        }
    } else {
        // 3-stage ring of direct-to-LDS tiles: two tiles in flight across the barrier (counted vmcnt + raw s_barrier;
        // __syncthreads() would drain the DMA queue).  Tile kt is consumed one barrier after the wait that retires it.
        constexpr int LPT = XR + WR;  // DMA instructions per tile per wave
        load_tile(0, 0);
        if (nkt > 1) load_tile(1, 1);
        for (int kt = 0; kt < nkt; ++kt) {
            if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < nkt) load_tile(kt + 2, (kt + 2) % 3);
            compute_tile(kt % 3);
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------------
    // lane holds D[co = cbase + i*16 + (lane>>4)*4 + r][pixel = mbase + j*16 + (lane&15)]
    const int cbase = tn * BN + wn * (BN / 2) + ((lane >> 4) << 2);
    const int mbase = tm * BM + wm * (BM / 2) + (lane & 15);
    const bool f32out = (p.flags & CY_CONV_BIAS_F32OUT) != 0;
    const bool accum = (p.flags & CY_CONV_ACCUM) != 0;

    if ((p.flags & CY_CONV_STATS) && !(p.dbg_nomma & 8)) {
        // per channel (sum, sumsq) of this block's BM pixels: 16-lane DPP row sums -> LDS [wm][2][BN] -> one coalesced
        // fp32 atomic per (channel, moment) into one of 64 bins (at most blocks/64 adds per address, no fold launch)
        if constexpr (NST != 2) __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            float sv = 0.f, qv = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const float v = (mbase + j * 16 < p.M) ? acc[i][j][r] : 0.f;
                    s += v;
                    q += v * v;
                }
                s = row16_sum(s);
                q = row16_sum(q);
                if ((lane & 3) == r) { sv = s; qv = q; }
            }
            // lanes 0..3 of each 16-lane row publish r = lane & 3
            if ((lane & 15) < 4) {
                const int cl = wn * (BN / 2) + i * 16 + ((lane >> 4) << 2) + (lane & 3);
                red[(wm * 2 + 0) * BN + cl] = sv;
                red[(wm * 2 + 1) * BN + cl] = qv;
            }
        }
        __syncthreads();
        float* srow = p.stats + (size_t)(lid & 63) * 2 * p.OC;
        for (int c = tid; c < 2 * BN; c += 256) {
            const int mom = c / BN, cl = c - mom * BN, co = tn * BN + cl;
            if (co < p.OC) atomicAdd(srow + mom * p.OC + co, red[mom * BN + cl] + red[(2 + mom) * BN + cl]);
        }
    }

    const bool sublattice = (p.oh_mul | p.ow_mul) != 1 || p.OHc != p.OH || p.OWc != p.OW;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int mj = mbase + j * 16;
        if (mj >= p.M || (p.dbg_nomma & 4)) continue;
        int m = mj;
        if (sublattice) {
            const int n = mj / ohw, rem = mj - n * ohw;
            const int ohc = rem / p.OWc;
            m = (n * p.OH + ohc * p.oh_mul + p.oh_off) * p.OW + (rem - ohc * p.OWc) * p.ow_mul + p.ow_off;
        }
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int co = cbase + i * 16;
            if (co >= p.OC) continue;
            f32x4 v = acc[i][j];
            if (p.flags & CY_CONV_AFFINE_ACT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = min(co + r, p.OC - 1);
                    const float z = v[r] * p.aff_scale[c] + p.aff_shift[c];
                    v[r] = p.act == CY_ACT_MISH ? mish_f<sizeof(T) == 2>(z) : (p.act == CY_ACT_LEAKY ? (z > 0.f ? z : 0.1f * z) : z);
                }
                if (p.res) {
                    const T* rp = reinterpret_cast<const T*>(p.res) + (size_t)m * p.ldres + co;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < p.OC) v[r] += (float)rp[r];
                }
            }
            if (f32out) {
                float* dst = reinterpret_cast<float*>(p.o) + (size_t)m * p.ldo + co;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co + r < p.OC) {
                        float t = v[r] + (p.bias ? p.bias[co + r] : 0.f);
                        if (accum) t += dst[r];
                        dst[r] = t;
                    }
            } else if (sizeof(T) == 2) {
                f16* dst = reinterpret_cast<f16*>(p.o) + (size_t)m * p.ldo + co;
                if (co + 3 < p.OC) {
                    if (accum) {
                        const f16x4 old = *reinterpret_cast<const f16x4*>(dst);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (float)old[r];
                    }
                    f16x4 h;
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[r] = (f16)v[r];
                    *reinterpret_cast<f16x4*>(dst) = h;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < p.OC) dst[r] = (f16)(v[r] + (accum ? (float)dst[r] : 0.f));
                }
            } else {
                float* dst = reinterpret_cast<float*>(p.o) + (size_t)m * p.ldo + co;
                if (co + 3 < p.OC) {
                    if (accum) {
                        const f32x4 old = *reinterpret_cast<const f32x4*>(dst);
                        v += old;
                    }
                    *reinterpret_cast<f32x4*>(dst) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < p.OC) dst[r] = v[r] + (accum ? dst[r] : 0.f);
                }
            }
        }
    }
}

template <typename T, int BM, int BN, bool GLDS, int NST>
int launch_v(const IgemmParams& p0, hipStream_t s) {
    IgemmParams p = p0;
    p.mtiles = (p.M + BM - 1) / BM;
    p.ntiles = (p.OC + BN - 1) / BN;
    static int lds_pad = -1;   // CY_IGEMM_LDS_PAD=bytes: occupancy experiments (forces fewer resident blocks per CU)
    if (lds_pad < 0) { const char* e = getenv("CY_IGEMM_LDS_PAD"); lds_pad = e ? atoi(e) : 0; }
    const int smem = NST * (BM + BN) * 128 + lds_pad;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<T, BM, BN, GLDS, NST>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_done = true;
    }
    hipLaunchKernelGGL((igemm_kernel<T, BM, BN, GLDS, NST>), dim3(p.mtiles * p.ntiles), dim3(256), smem, s, p);
    CY_LAUNCH_CHECK();
    return 0;
}

// CY_IGEMM_GLDS=0 selects the register-staged double buffer (kept for A/B measurements on the 64/128-pixel tiles);
// default is the direct-to-LDS double buffer.  A 3-stage DMA ring (NST = 3) measured slower at equal LDS footprint.
inline int glds_mode() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("CY_IGEMM_GLDS");
        v = e ? atoi(e) : 1;
    }
    return v;
}

template <typename T, int BM, int BN>
int launch(const IgemmParams& p, hipStream_t s) {
    if constexpr (BM == 128 || BM == 64) {
        if (glds_mode() == 0) return launch_v<T, BM, BN, false, 2>(p, s);
    }
    return launch_v<T, BM, BN, true, 2>(p, s);
}

// Tile choice.  Channel tile = min(128, OC rounded up to 32).  Pixel tile: blocks run in rounds of 256 * blocks-per-CU
// (LDS-limited) and a K step costs ~1 us per resident block almost independently of the tile area (the double-buffered
// loop is latency-bound; tools/tile_sweep*.sh), so
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
    } else if (bn >= 64 && tile_rounds(M, OC, 192, bn) < tile_rounds(M, OC, 128, bn)) {
        bm = 192;
    }
}

template <typename T>
int dispatch(const IgemmParams& p, hipStream_t s) {
    int bm, bn;
    pick_tile(p.M, p.OC, p.ntaps * p.GC, (int)sizeof(T), bm, bn);
    {   // CY_IGEMM_TILE=BMxBN forces a tile (tuning experiments)
        static int fbm = -1, fbn = -1;
        if (fbm < 0) {
            const char* e = getenv("CY_IGEMM_TILE");
            fbm = fbn = 0;
            if (e) sscanf(e, "%dx%d", &fbm, &fbn);
        }
        if (fbm > 0) { bm = fbm; bn = fbn; }
    }
#define CY_TILE(BM_, BN_) \
    if (bm == BM_ && bn == BN_) return launch<T, BM_, BN_>(p, s);
    CY_TILE(128, 128) CY_TILE(128, 64) CY_TILE(128, 32) CY_TILE(64, 128) CY_TILE(64, 64) CY_TILE(64, 32)
    CY_TILE(192, 128) CY_TILE(192, 64) CY_TILE(160, 128) CY_TILE(160, 64) CY_TILE(96, 128) CY_TILE(96, 64) CY_TILE(256, 64)
    CY_TILE(256, 128)
#undef CY_TILE
    return CY_ERR_ARG;
}

}  // namespace

extern "C" int cy_conv_stats_rows(int M, int OC) {
    (void)M; (void)OC;
    return 64;
}

static int conv_igemm_impl(const void* g, int N, int GH, int GW, int GC, int ldg, const void* w, int wrows, void* out,
                           int OH, int OW, int OC, int ldo, int ks, int stride, int pad, int dtype, int flags,
                           const float* bias, float* stats_part, int* stats_rows_host, const float* aff_scale,
                           const float* aff_shift, int act, const void* res, int ldres, cy_stream_t s) {
    const int ch = dtype == CY_F16 ? 8 : 4;
    if (!g || !w || !out || (dtype != CY_F16 && dtype != CY_F32)) return CY_ERR_ARG;
    if ((ks != 1 && ks != 3) || (stride != 1 && stride != 2) || GC % ch || ldg % ch) return CY_ERR_ARG;
    if ((flags & CY_CONV_STATS) && !stats_part) return CY_ERR_ARG;
    if (!(flags & CY_CONV_BIAS_F32OUT) && (ldo % 4)) return CY_ERR_ARG;
    IgemmParams p;
    p.g = (const unsigned char*)g; p.w = (const unsigned char*)w; p.o = (unsigned char*)out;
    p.bias = bias; p.stats = stats_part;
    {
        static int nomma = -1;
        if (nomma < 0) { const char* e = getenv("CY_IGEMM_NOMMA"); nomma = e ? atoi(e) : 0; }
        p.dbg_nomma = nomma;
    }
    p.aff_scale = aff_scale; p.aff_shift = aff_shift; p.act = act; p.res = (const unsigned char*)res; p.ldres = ldres;
    if ((flags & CY_CONV_AFFINE_ACT) && (!aff_scale || !aff_shift || (flags & (CY_CONV_STATS | CY_CONV_TRANSPOSED)))) return CY_ERR_ARG;
    p.N = N; p.GH = GH; p.GW = GW; p.GC = GC; p.ldg = ldg;
    p.OH = OH; p.OW = OW; p.OC = OC; p.ldo = ldo;
    p.ks = ks; p.stride = stride; p.pad = pad; p.transposed = (flags & CY_CONV_TRANSPOSED) ? 1 : 0;
    p.K = ks * ks * GC; p.M = N * OH * OW; p.wrows = wrows; p.flags = flags;
    p.mtiles = p.ntiles = 0;
    if (p.M <= 0 || OC <= 0) return CY_ERR_ARG;
    if (stats_rows_host) *stats_rows_host = cy_conv_stats_rows(p.M, OC);
    const size_t esz = dtype == CY_F16 ? 2 : 4;
    const size_t gb = (((size_t)N * GH * GW - 1) * ldg + GC) * esz, wb = (size_t)wrows * p.K * esz;
    if (gb >= 0xFFFFFF00ull || wb >= 0xFFFFFF00ull) return CY_ERR_ARG;  // 32-bit buffer offsets
    p.g_bytes = (unsigned)gb; p.w_bytes = (unsigned)wb;
    p.OHc = OH; p.OWc = OW; p.oh_mul = p.ow_mul = 1; p.oh_off = p.ow_off = 0;
    p.ntaps = ks * ks; p.kh_pack = p.kw_pack = 0;
    for (int t = 0; t < ks * ks; ++t) {
        p.kh_pack |= (unsigned)(t / ks) << (2 * t);
        p.kw_pack |= (unsigned)(t % ks) << (2 * t);
    }
    if (!(p.transposed && stride == 2))
        return dtype == CY_F16 ? dispatch<f16>(p, cy_s(s)) : dispatch<float>(p, cy_s(s));
    // stride-2 dgrad: an input-gradient pixel only sees the taps with (o + pad - k) even.  Four launches, one per
    // (row, column) parity class, each over its own taps: 9 tap-visits in total instead of 36.
    for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw) {
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
            const int rc = dtype == CY_F16 ? dispatch<f16>(q, cy_s(s)) : dispatch<float>(q, cy_s(s));
            if (rc) return rc;
        }
    return 0;
}

extern "C" int cy_conv_igemm(const void* g, int N, int GH, int GW, int GC, int ldg, const void* w, int wrows,
                             void* out, int OH, int OW, int OC, int ldo, int ks, int stride, int pad, int dtype,
                             int flags, const float* bias, float* stats_part, int* stats_rows_host, cy_stream_t s) {
    CY_ENTER();
    if (flags & CY_CONV_AFFINE_ACT) return CY_ERR_ARG;
    return conv_igemm_impl(g, N, GH, GW, GC, ldg, w, wrows, out, OH, OW, OC, ldo, ks, stride, pad, dtype, flags, bias,
                           stats_part, stats_rows_host, nullptr, nullptr, 0, nullptr, 0, s);
}

extern "C" int cy_conv_bn_act_eval(const void* g, int N, int GH, int GW, int GC, int ldg, const void* w, int wrows,
                                   void* out, int OH, int OW, int OC, int ldo, int ks, int stride, int pad, int dtype,
                                   const float* scale, const float* shift, int act, const void* res, int ldres,
                                   cy_stream_t s) {
    CY_ENTER();
    return conv_igemm_impl(g, N, GH, GW, GC, ldg, w, wrows, out, OH, OW, OC, ldo, ks, stride, pad, dtype,
                           CY_CONV_AFFINE_ACT, nullptr, nullptr, nullptr, scale, shift, act, res, ldres, s);
}
