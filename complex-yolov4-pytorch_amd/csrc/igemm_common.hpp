// Shared pieces of the implicit-GEMM convolution kernels (conv_igemm.hip: 4-wave double-buffered gather kernels, every
// dtype and shape; conv_pipe.hip: 8-wave 3-stage pipelined kernel for the 16-bit modes): launch parameters, the 16x16
// MFMA fragment helpers and conv_igemm.hip's epilogue (BN batch statistics, eval-mode affine + activation, packed NHWC
// stores).
#pragma once
#include <type_traits>

#include "common.hpp"

namespace cyk {

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
    // CY_CONV_BNBWD_SUMS epilogue (conv_pipe.hip): res / ldres = the producer layer's pre-BN tensor, aff_scale / aff_shift
    // its BN affine, bn_mean / bn_invstd its batch statistics; the sums go to `stats`
    const float* bn_mean;
    const float* bn_invstd;
    unsigned x_bias;              // fast / pipelined kernels: bytes the gather descriptor's base sits below g (>= any negative row offset)
    int bm_eff;                   // conv_pipe.hip: pixels per tile actually used (<= the kernel's tile capacity)
    int stat_det;                 // statistics table: 0 = CY_STAT_BINS bins shared by the blocks (atomics, bin = tile % CY_STAT_BINS);
                                  // 1 = one row per pixel tile (one add per address onto zero: run-to-run deterministic)
    // Stride-2 dgrad as ONE launch: the four (row, column) parity classes of the input-gradient lattice share the grid, pixel
    // tile index & 3 = class (heaviest first: 4, 2, 2, 1 taps for 3x3 / pad 1), every class over M pixels of its own OHc x OWc
    // sub-lattice (the host merges only when the four sub-lattices are congruent, i.e. OH and OW even).  1: an ordinary launch.
    int ncls;
    // CY_CONV_BN_FUSED epilogue (conv_pipe.hip, cy_conv_bn_act_train): o / ldo = the pre-BN tensor, o2 / ldo2 = the activated
    // output, res / ldres = the shortcut operand, stats = the (sum, sumsq) bins; BatchNorm parameters and state below
    unsigned char* o2;
    int ldo2;
    const float* bn_gamma;
    const float* bn_beta;
    float* bn_rmean;
    float* bn_rvar;
    long long* bn_nbt;
    float bn_momentum, bn_eps;
    float* bn_vec;
    float* bn_zero;
    int bn_zero_n;
    int* ticket;
    int slab_rows;                // conv3x3_slab_kernel (conv_pipe.hip): rows of one input slab (tile pixels + 2 GW + 2, a multiple of 8)
};

// What a block needs to know about ITS pixel lattice and taps: the launch's own (ordinary launches, one parity class per
// launch) or, in a merged stride-2 dgrad, those of the class its tile index selects -- derived from the class number with a
// few scalar instructions (no per-class tables in the kernel arguments: a dynamically indexed argument array would put the
// whole parameter block into scratch memory).
struct TapSet {
    int oh_off, ow_off, ntaps;
    unsigned kh_pack, kw_pack;
};

// -> the block's pixel-tile index inside its class
__device__ __forceinline__ int select_class(const IgemmParams& p, int tm, TapSet& c) {
    c.oh_off = p.oh_off; c.ow_off = p.ow_off; c.ntaps = p.ntaps; c.kh_pack = p.kh_pack; c.kw_pack = p.kw_pack;
    if (p.ncls <= 1) return tm;
    const int k = tm & 3, ph = 1 - (k >> 1), pw = 1 - (k & 1);
    c.oh_off = ph; c.ow_off = pw; c.ntaps = 0; c.kh_pack = c.kw_pack = 0u;
    for (int kh = 0; kh < p.ks; ++kh) {
        if ((ph + p.pad - kh) & 1) continue;
        for (int kw = 0; kw < p.ks; ++kw) {
            if ((pw + p.pad - kw) & 1) continue;
            c.kh_pack |= (unsigned)kh << (2 * c.ntaps);
            c.kw_pack |= (unsigned)kw << (2 * c.ntaps);
            ++c.ntaps;
        }
    }
    return tm >> 2;
}

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
struct Mma<bf16> {
    static constexpr int KSTEPS = 2;
    typedef bf16x8 frag;
    __device__ static __forceinline__ frag load(const unsigned char* row_ptr, int kk, int lane) {
        const int c = (kk * 4 + (lane >> 4)) ^ (lane & 7);
        return *reinterpret_cast<const frag*>(row_ptr + (c << 4));
    }
    __device__ static __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
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

// ---- 32 x 32 x 16 fragments (conv_pipe.hip, conv_direct.hip) ------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <typename T>
struct Mma32;
template <>
struct Mma32<f16> {
    typedef f16x8 frag;
    __device__ static __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
template <>
struct Mma32<bf16> {
    typedef bf16x8 frag;
    __device__ static __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};

// sum over the 32 lanes of a half wave that share lane >> 5 (every lane of the half ends up with the total)
__device__ __forceinline__ float half32_sum(float v) {
    v = row16_sum(v);
    return v + __shfl_xor(v, 16, 64);
}

// Epilogue of a BN-channel x BM-pixel block tile held as 16x16 accumulator fragments:
// lane holds D[co = cbase + i*16 + (lane>>4)*4 + r][pixel = mbase + j*16 + (lane&15)].
// NW waves = 2 over channels x NW/2 over pixels.  smem: the block's LDS (free for reuse; SYNC_FIRST adds the barrier
// that makes it so when the main loop does not end with one).
template <typename T, int BM, int BN, int NW, bool SYNC_FIRST>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& p, const TapSet& cls, f32x4 (&acc)[BN / 32][BM / (8 * NW)], int tm,
                                               int tn, int lid, unsigned char* smem) {
    constexpr int NT = NW * 64, WMW = NW / 2, TI = BN / 32, TJ = BM / (16 * WMW);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 1, wm = wave >> 1;
    const int ohw = p.OHc * p.OWc;
    // ---- epilogue ---------------------------------------------------------------------------------
    // lane holds D[co = cbase + i*16 + (lane>>4)*4 + r][pixel = mbase + j*16 + (lane&15)]
    const int cbase = tn * BN + wn * (BN / 2) + ((lane >> 4) << 2);
    const int mbase = tm * BM + wm * (BM / WMW) + (lane & 15);
    const bool f32out = (p.flags & CY_CONV_BIAS_F32OUT) != 0;
    const bool accum = (p.flags & CY_CONV_ACCUM) != 0;

    if (p.flags & CY_CONV_STATS) {
        // per channel (sum, sumsq) of this block's BM pixels: 16-lane DPP row sums -> LDS [wm][2][BN] -> one coalesced
        // fp32 atomic per (channel, moment) into one of 64 bins (at most blocks/64 adds per address, no fold launch)
        if constexpr (SYNC_FIRST) __syncthreads();
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
        float* srow = p.stats + (size_t)(p.stat_det ? tm : (lid & (CY_STAT_BINS - 1))) * 2 * p.OC;
        for (int c = tid; c < 2 * BN; c += NT) {
            const int mom = c / BN, cl = c - mom * BN, co = tn * BN + cl;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WMW; ++w) t += red[(2 * w + mom) * BN + cl];
            if (co < p.OC) atomicAdd(srow + mom * p.OC + co, t);
        }
    }

    const bool sublattice = (p.oh_mul | p.ow_mul) != 1 || p.OHc != p.OH || p.OWc != p.OW;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int mj = mbase + j * 16;
        if (mj >= p.M) continue;
        int m = mj;
        if (sublattice) {
            const int n = mj / ohw, rem = mj - n * ohw;
            const int ohc = rem / p.OWc;
            m = (n * p.OH + ohc * p.oh_mul + cls.oh_off) * p.OW + (rem - ohc * p.OWc) * p.ow_mul + cls.ow_off;
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
            } else if constexpr (sizeof(T) == 2) {
                typedef T tx4 __attribute__((ext_vector_type(4)));
                T* dst = reinterpret_cast<T*>(p.o) + (size_t)m * p.ldo + co;
                if (co + 3 < p.OC) {
                    if (accum) {
                        const tx4 old = *reinterpret_cast<const tx4*>(dst);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (float)old[r];
                    }
                    tx4 h;
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[r] = (T)v[r];
                    *reinterpret_cast<tx4*>(dst) = h;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < p.OC) dst[r] = (T)(v[r] + (accum ? (float)dst[r] : 0.f));
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

}  // namespace cyk

// conv_pipe.hip: launches the pipelined kernel when the shape qualifies (*used = 1), else leaves the launch to
// conv_igemm.hip (*used = 0).
int cy_pipe_try(const cyk::IgemmParams& p, int dtype, hipStream_t s, int* used);
// conv_direct.hip: the small-Cin 3x3 forward layers (3 -> 32, 32 -> 64) as a direct convolution over an LDS-resident input
// patch (*used = 1), everything else (*used = 0) stays with the implicit-GEMM kernels.
int cy_direct_try(const cyk::IgemmParams& p, int dtype, hipStream_t s, int* used);
