// 3x3 / stride 1 / pad 1 implicit-GEMM convolution (forward and dgrad) with the input patch resident in LDS.
//
// Why a second kernel: the generic gather kernel (conv_igemm.hip) re-fetches the activation tile for each of the nine
// taps, so per 128x128x64 MAC step it moves 32 KB through the CU's vector-memory path (TA / L1, <= 64 B/clk) for 512
// clocks of MFMA work -- measured, that path and not the MFMA pipe bounds it (skipping the MFMAs removes only 25 % of
// the time).  Here a block owns BM consecutive output pixels of the flattened (n, oh, ow) raster and keeps the input
// rows it can touch -- the same raster range extended by W + 1 pixels on either side -- in LDS for one 64-channel
// (128-byte) slice: the nine taps are nine shifted views of that patch.  Per (channel slice, tap) step only the BN x
// 128-byte weight tile and 1/9 of the NEXT slice's patch are fetched: about 3x fewer bytes per MAC at BM = 256.
//
// Zero padding cannot come from the DMA's out-of-range zero fill any more (a patch row is shared by taps that see it
// as inside and as outside the image), so every output pixel carries a 4-bit edge code (top, bottom, left, right) and
// the B fragments of the taps that leave the image are replaced by zeros after the LDS read.
//
// LDS: [patch 0][patch 1][weights 0][weights 1]; patch = ceil((BM + 2W + 2) / 8) pieces of 8 rows x 128 B, rows
// XOR-swizzled by (row & 7) exactly like the generic kernel's tiles (the swizzle key of a shifted view is the patch
// row's, not the lane's).  One barrier per step; dgrad = same kernel with mirrored taps over the [Cin][tap, Cout] pack.
#include <stdio.h>
#include <stdlib.h>

#include "igemm_common.hpp"

namespace {
using namespace cyk;

template <typename T>
struct FragOps;
template <>
struct FragOps<f16> {
    typedef Mma<f16>::frag frag;
    __device__ static __forceinline__ frag load_sw(const unsigned char* row_ptr, int kk, int lane, int sw) {
        const int c = (kk * 4 + (lane >> 4)) ^ sw;
        return *reinterpret_cast<const frag*>(row_ptr + (c << 4));
    }
    __device__ static __forceinline__ frag zero_if(frag v, bool z) {
        const frag zero = {};
        return z ? zero : v;
    }
};
template <>
struct FragOps<float> {
    typedef float frag;
    __device__ static __forceinline__ frag load_sw(const unsigned char* row_ptr, int kk, int lane, int sw) {
        const int c = kk ^ sw;
        return *reinterpret_cast<const float*>(row_ptr + (c << 4) + ((lane >> 4) << 2));
    }
    __device__ static __forceinline__ frag zero_if(frag v, bool z) { return z ? 0.f : v; }
};

template <typename T, int BM, int BN, int NW>
__global__ void __launch_bounds__(NW * 64) halo3x3_kernel(const IgemmParams p) {
    constexpr int CH = Elem<T>::CH;
    constexpr int BK = 8 * CH;                 // channels per 128-byte row
    constexpr int NT = NW * 64, RS = NW * 8;
    constexpr int WMW = NW / 2;
    constexpr int WR = BN / RS;
    constexpr int TI = BN / 32, TJ = BM / (16 * WMW);
    static_assert(BN % RS == 0 && BM % (16 * WMW) == 0 && BM <= NT, "tile / wave layout");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 1, wm = wave >> 1;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, slot = bid >> 3;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int tn = lid % p.ntiles, tm = lid / p.ntiles;

    const int W = p.GW, H = p.GH;
    const int m0 = tm * BM;
    const int g0 = m0 - W - 1;                 // flattened input pixel held by patch row 0 (may be negative)
    const int sgn = p.transposed ? -1 : 1;     // dgrad reads x[o + 1 - k], forward x[o - 1 + k]

    unsigned char* const xbuf0 = smem;
    unsigned char* const wbuf0 = smem + 2 * p.halo_xbuf;

    // ---- edge codes: bit t of inv[j] = tap t of this lane's pixel j leaves the image ---------------------------------
    unsigned inv[TJ];
    {
        unsigned short* codes = reinterpret_cast<unsigned short*>(smem);
        if (tid < BM) {
            const int m = m0 + tid;
            unsigned code = 0x1FFu;
            if (m < p.M) {
                const int hw = H * W;
                const int n = m / hw, rem = m - n * hw;
                const int oh = rem / W, ow = rem - oh * W;
                code = 0u;
                for (int t = 0; t < 9; ++t) {
                    const int dh = sgn * (t / 3 - 1), dw = sgn * (t % 3 - 1);
                    const bool out = (unsigned)(oh + dh) >= (unsigned)H || (unsigned)(ow + dw) >= (unsigned)W;
                    code |= (out ? 1u : 0u) << t;
                }
            }
            codes[tid] = (unsigned short)code;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TJ; ++j) inv[j] = codes[wm * (BM / WMW) + j * 16 + (lane & 15)];
        __syncthreads();   // the patch overlays the codes
    }

    const auto rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)p.g, 0, p.g_bytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFF00u;

    // direct-to-LDS pieces: lane l fills physical chunk l & 7 of row l >> 3, i.e. fetches logical chunk (l&7)^(row&7)
    const int lrow = lane >> 3;
    const unsigned lchunk_bytes = (unsigned)(((lane & 7) ^ lrow) * CH) * (unsigned)sizeof(T);
    const unsigned pix_bytes = (unsigned)p.ldg * (unsigned)sizeof(T);
    auto load_patch_piece = [&](int q, int cc, int buf) {
        const int g = g0 + q * 8 + lrow;
        const bool ok = (unsigned)g < (unsigned)p.M;
        const unsigned off = ok ? (unsigned)g * pix_bytes + (unsigned)(cc * BK) * (unsigned)sizeof(T) + lchunk_bytes : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rs_g, (__attribute__((address_space(3))) void*)(xbuf0 + buf * p.halo_xbuf + q * 1024), 16, off, 0, 0, 0);
    };
    const unsigned w_row0 = (unsigned)(tn * BN + wave * 8 + lrow) * (unsigned)p.K * (unsigned)sizeof(T);
    const unsigned w_rstep = (unsigned)RS * (unsigned)p.K * (unsigned)sizeof(T);
    auto load_weights = [&](int cc, int t, int stage) {
        unsigned char* ws_w = wbuf0 + stage * (BN * 128) + wave_u * (8 * 128);
        const unsigned koff = (unsigned)(t * p.GC + cc * BK) * (unsigned)sizeof(T) + lchunk_bytes;
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            const bool ok = tn * BN + wave * 8 + lrow + RS * i < p.wrows;
            const unsigned off = ok ? w_row0 + i * w_rstep + koff : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(ws_w + i * (RS * 128)),
                                                     16, off, 0, 0, 0);
        }
    };

    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nc = p.GC / BK;
    const int npieces = p.halo_pieces;
    const int local0 = wm * (BM / WMW) + (lane & 15) + W + 1;   // patch row of this lane's pixel 0 under the centre tap

    for (int q = wave_u; q < npieces; q += NW) load_patch_piece(q, 0, 0);
    load_weights(0, 0, 0);
    __syncthreads();

    int cc = 0, t = 0, stage = 0;
    const int nsteps = nc * 9;
    for (int s = 0; s < nsteps; ++s) {
        // ---- prefetch: weights of the next step, one ninth of the next channel slice's patch ------------------------
        if (s + 1 < nsteps && !(p.dbg_nomma & 32)) {
            const int t1 = t == 8 ? 0 : t + 1, cc1 = t == 8 ? cc + 1 : cc;
            load_weights(cc1, t1, stage ^ 1);
        }
        if (cc + 1 < nc && !(p.dbg_nomma & 32))
            for (int q = t * NW + wave_u; q < npieces; q += 9 * NW) load_patch_piece(q, cc + 1, (cc + 1) & 1);

        // ---- MFMA over this (slice, tap) -------------------------------------------------------------------------
        if (!(p.dbg_nomma & 1)) {
            const unsigned char* ws = wbuf0 + stage * (BN * 128);
            const unsigned char* wrow = ws + (wn * (BN / 2) + (lane & 15)) * 128;
            const int pr0 = local0 + sgn * ((t / 3 - 1) * W + (t % 3 - 1));
            const int sw = pr0 & 7;
            const unsigned char* xrow = xbuf0 + (cc & 1) * p.halo_xbuf + pr0 * 128;
            bool z[TJ];
#pragma unroll
            for (int j = 0; j < TJ; ++j) z[j] = (inv[j] >> t) & 1u;
#pragma unroll
            for (int kk = 0; kk < Mma<T>::KSTEPS; ++kk) {
                typename Mma<T>::frag a[TI], b[TJ];
#pragma unroll
                for (int i = 0; i < TI; ++i) a[i] = Mma<T>::load(wrow + i * 16 * 128, kk, lane);
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    b[j] = FragOps<T>::zero_if(FragOps<T>::load_sw(xrow + j * 16 * 128, kk, lane, sw), z[j]);
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) acc[i][j] = Mma<T>::mma(a[i], b[j], acc[i][j]);
            }
        }
        __syncthreads();
        stage ^= 1;
        if (++t == 9) { t = 0; ++cc; }
    }

    igemm_epilogue<T, BM, BN, NW, false>(p, acc, tm, tn, lid, smem);
}

template <typename T, int BM, int BN, int NW>
int halo_launch_v(const IgemmParams& p0, hipStream_t s) {
    IgemmParams p = p0;
    p.mtiles = (p.M + BM - 1) / BM;
    p.ntiles = (p.OC + BN - 1) / BN;
    p.halo_pieces = (BM + 2 * p.GW + 2 + 7) / 8;
    p.halo_xbuf = p.halo_pieces * 1024;
    const int smem = 2 * p.halo_xbuf + 2 * BN * 128;
    static int attr_max = 0;
    if (smem > attr_max) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&halo3x3_kernel<T, BM, BN, NW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_max = 160 * 1024;
    }
    hipLaunchKernelGGL((halo3x3_kernel<T, BM, BN, NW>), dim3(p.mtiles * p.ntiles), dim3(NW * 64), smem, s, p);
    CY_LAUNCH_CHECK();
    return 0;
}

inline int halo_smem(int bm, int bn, int W) { return 2 * ((bm + 2 * W + 2 + 7) / 8) * 1024 + 2 * bn * 128; }

template <typename T>
int halo_dispatch(const IgemmParams& p, int bm, int bn, hipStream_t s) {
    if (bm == 256 && bn == 128) return halo_launch_v<T, 256, 128, 8>(p, s);
    if (bm == 256 && bn == 64) return halo_launch_v<T, 256, 64, 8>(p, s);
    if (bm == 128 && bn == 128) return halo_launch_v<T, 128, 128, 4>(p, s);
    if (bm == 128 && bn == 64) return halo_launch_v<T, 128, 64, 4>(p, s);
    return CY_ERR_ARG;
}

}  // namespace

static int64_t g_halo_launches = 0;

extern "C" int64_t cy_halo_launches(void) { return g_halo_launches; }

// Tile choice for the halo kernel; *used = 0 leaves the launch to the generic kernel.  Environment (read per call, so
// tests can switch it): CY_HALO=1 enables the kernel (default off: measured 10-15 % slower than the generic kernel on
// v4's shapes), CY_HALO_TILE=BMxBN forces a tile, CY_HALO_MINBLOCKS=n overrides the smallest grid.
int cy_halo3x3_try(const cyk::IgemmParams& p, int dtype, hipStream_t s, int* used) {
    *used = 0;
    int fbm = 0, fbn = 0, minblocks = 192;
    const char* en = getenv("CY_HALO");
    if (!en || !atoi(en)) return 0;
    if (const char* t = getenv("CY_HALO_TILE")) sscanf(t, "%dx%d", &fbm, &fbn);
    if (const char* mb = getenv("CY_HALO_MINBLOCKS")) minblocks = atoi(mb);
    const int bk = dtype == CY_F16 ? 64 : 32;
    if (p.ks != 3 || p.stride != 1 || p.pad != 1 || p.GH != p.OH || p.GW != p.OW || p.GC % bk) return 0;
    if (p.ntaps != 9 || p.OHc != p.OH || p.OWc != p.OW || (p.oh_mul | p.ow_mul) != 1) return 0;
    if (p.OC <= 32) return 0;
    int bn = p.OC > 64 ? 128 : 64, bm = 256;
    auto blocks = [&](int m) { return (long)((p.M + m - 1) / m) * ((p.OC + bn - 1) / bn); };
    auto fits = [&](int m) { return halo_smem(m, bn, p.GW) <= 160 * 1024; };
    if (fbm) {
        bm = fbm; bn = fbn;
    } else {
        if (!fits(256) || blocks(256) < minblocks) bm = 128;
        if (bm == 128 && (!fits(128) || blocks(128) < minblocks)) return 0;
    }
    if (!fits(bm)) return 0;
    const int rc = dtype == CY_F16 ? halo_dispatch<f16>(p, bm, bn, s) : halo_dispatch<float>(p, bm, bn, s);
    if (rc == 0) { *used = 1; ++g_halo_launches; }
    return rc;
}
