// Pipelined implicit-GEMM convolution (forward and dgrad) for the 16-bit storage modes: the kernel the compute-bound
// and the large HBM-bound layers of complex_yolov4.cfg run on (reference work unit: darknet2pytorch.py:247-278).
//
// Same GEMM view and the same gather scheme as conv_igemm.hip's fast path (D[co][pixel] = sum_k W[co][k] * X[pixel][k],
// tap and channel offset wave-uniform per K step, out-of-range buffer offsets = zero fill), but built around what the
// 2-barrier double buffer cannot do:
//   * 512 threads = 8 waves (2 over channels x 4 over pixels), ONE block per CU, tile BN channels x BM pixels with
//     BM = 256: 25 % fewer bytes through the CU's load path per MAC than two 128 x 128 blocks;
//   * a 3-stage LDS ring filled by buffer_load ... lds with COUNTED s_waitcnt vmcnt(N) and a raw s_barrier: two K tiles
//     stay in flight across every barrier (a __syncthreads() would drain the DMA queue each step);
//   * v_mfma_f32_32x32x16 fragments (half the MFMA issue slots of 16x16x32, 2382 vs 2075 TF ubench ceiling);
//   * LDS rows are 128 B (64 halfs), 16-byte chunk index XOR (row >> 1) & 7 -- applied on the DMA SOURCE side, the
//     DMA image being lane-linear -- which is conflict-free for the 32-row ds_read_b128 fragment reads;
//   * epilogue through LDS: every wave transposes its 64-channel x 64-pixel accumulator block to pixel-major rows and
//     stores 16 B per lane, 128 contiguous bytes per pixel (the MFMA layout itself only yields 16-byte segments).
// BatchNorm statistics, gradient fan-in (ACCUM), the eval-mode affine + activation (+ shortcut) epilogue and the
// stride-2 dgrad parity classes behave exactly as in conv_igemm.hip (same IgemmParams, same stats table).
#include <stdio.h>
#include <stdlib.h>

#include "igemm_common.hpp"

// CY_ABL: ablation switches of conv3x3_slabk_kernel for tools/abl_build.sh (0 in every shipped build: the hooks compile to nothing)
#ifndef CY_ABL
#define CY_ABL 0
#endif

namespace {
using namespace cyk;

// The epilogue shared by the kernels of this file: statistics, the eval-mode / two-phase BatchNorm + activation, gradient fan-in,
// BN-backward sums, LDS-transposed or direct stores.  acc: this wave's TI x TJ 32 x 32 accumulator fragments; smem: the block's LDS
// (LDS_BYTES of it free for reuse -- every wave is past its last read of the main loop's tiles); orow: byte offset of every tile
// row's output pixel (0xFFFFFFFF: no pixel), kept outside the reused region.
template <typename T, int BM, int BN, int WN, bool EPI_LDS, int LDS_BYTES, bool BNF = true>
__device__ __forceinline__ void pipe_epilogue(const IgemmParams& p, f32x16 (&acc)[BN / (32 * WN)][BM / (32 * (8 / WN))], unsigned char* smem,
                                              const unsigned* orow, const int tn, const int tm, const int lid) {
    constexpr int WM = 8 / WN;
    constexpr int TI = BN / (32 * WN), TJ = BM / (32 * WM);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = (wave & 7) % WN, wm = (wave & 7) / WN;
    // ---- epilogue -----------------------------------------------------------------------------------------------------
    // lane holds D[co = cw + i*32 + 8*g + 4*(lane>>5) + r][pixel row = pw + j*32 + (lane&31)], acc register 4*g + r
    const int half = lane >> 5;
    const int cw = wn * (BN / WN);         // first channel of this wave inside the tile
    const int pw = wm * (BM / WM);         // first pixel row of this wave inside the tile
    const int co_w = tn * BN + cw;

    if (p.flags & CY_CONV_STATS) {
        float* red = reinterpret_cast<float*>(smem);        // [WM][2][BN]
        bool rowok[TJ];
#pragma unroll
        for (int j = 0; j < TJ; ++j) rowok[j] = orow[pw + j * 32 + (lane & 31)] != 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            float sv = 0.f, qv = 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const float v = rowok[j] ? acc[i][j][t] : 0.f;
                    s += v;
                    q += v * v;
                }
                s = half32_sum(s);
                q = half32_sum(q);
                if ((lane & 31) == t) { sv = s; qv = q; }
            }
            if ((lane & 31) < 16) {     // lane t of each half publishes accumulator register t = 4 g + r
                const int t = lane & 31;
                const int cl = cw + i * 32 + 8 * (t >> 2) + 4 * half + (t & 3);
                red[(wm * 2 + 0) * BN + cl] = sv;
                red[(wm * 2 + 1) * BN + cl] = qv;
            }
        }
        __syncthreads();
        float* srow = p.stats + (size_t)(p.stat_det ? tm : (lid & (CY_STAT_BINS - 1))) * 2 * p.OC;
        for (int c = tid; c < 2 * BN; c += 512) {
            const int mom = c / BN, cl = c - mom * BN, co = tn * BN + cl;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) t += red[(2 * w + mom) * BN + cl];
            if (co < p.OC) atomicAdd(srow + mom * p.OC + co, t);
        }
        __syncthreads();
    }

    // CY_CONV_BN_FUSED (cy_conv_bn_act_train): BatchNorm with BATCH statistics + activation in this launch.  Phase 1 ends
    // here: this block's sums are on their way into the bins.  Wait until they have been performed (vmcnt counts atomics on
    // gfx9), then ARRIVE at the grid's ticket; the pre-BN tile is stored below while the other blocks arrive.
    const bool bnf = BNF && (p.flags & CY_CONV_BN_FUSED) != 0;      // (BNF = false: a kernel without the two-phase epilogue)
    const int nblk_all = p.mtiles * p.ntiles;
    if (bnf) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(p.ticket, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }

    if (p.flags & CY_CONV_AFFINE_ACT) {
        typedef T rx4 __attribute__((ext_vector_type(4)));
        const T* resrow[TJ];     // the shortcut operand has the output's pixel indexing (its own channel stride)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const unsigned ob = orow[pw + j * 32 + (lane & 31)];
            resrow[j] = (p.res && ob != 0xFFFFFFFFu)
                            ? reinterpret_cast<const T*>(p.res) + (size_t)(ob / ((unsigned)p.ldo * (unsigned)sizeof(T))) * p.ldres
                            : nullptr;
        }
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = co_w + i * 32 + 8 * g + 4 * half;
                float sc[4], sf[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = min(co + r, p.OC - 1);
                    sc[r] = p.aff_scale[c];
                    sf[r] = p.aff_shift[c];
                }
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float z = acc[i][j][4 * g + r] * sc[r] + sf[r];
                        const float zm = mish_f<true>(z), zl = z > 0.f ? z : 0.1f * z;
                        acc[i][j][4 * g + r] = p.act == CY_ACT_MISH ? zm : (p.act == CY_ACT_LEAKY ? zl : z);   // selects, no branches
                    }
                    if (resrow[j] && co + 3 < p.OC) {
                        const rx4 rv = *reinterpret_cast<const rx4*>(resrow[j] + co);
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[i][j][4 * g + r] += (float)rv[r];
                    }
                }
            }
    }

    const bool accum = (p.flags & CY_CONV_ACCUM) != 0;
    typedef T tx4 __attribute__((ext_vector_type(4)));
    typedef T tx8 __attribute__((ext_vector_type(8)));
    if constexpr (EPI_LDS) {
        // wave-private [BM/WM pixel rows][BN/WN channels] tile, 16-byte chunk index XOR (row & 7)
        constexpr int ROWB = (BN / WN) * 2;      // bytes per pixel row of the wave tile (128 for 64 channels)
        constexpr int CPR = ROWB / 16;           // 16-byte chunks per row
        constexpr int WROWS = BM / WM;
        unsigned char* wt = smem + wave * (WROWS * ROWB);
        if (bnf) {
            // (its own copy of the plain store loop, and a return: the accumulators stay live across the grid wait here, which
            // the general path below -- whose fan-in / sums prefetches reuse their registers -- must not pay for)
            constexpr int RPI2 = 64 / CPR, NIT2 = WROWS / RPI2;
            const unsigned opix = (unsigned)p.ldo * (unsigned)sizeof(T);      // bytes per pixel row of the pre-BN tensor
            auto store_rows = [&](unsigned char* base, unsigned pixbytes) {     // LDS tile -> 16-byte stores, 128 B per pixel
#pragma unroll
                for (int t = 0; t < NIT2; ++t) {
                    const int row = t * RPI2 + lane / CPR, c = lane % CPR;
                    const unsigned ob = orow[pw + row];
                    const int co = co_w + c * 8;
                    const tx8 v = *reinterpret_cast<const tx8*>(wt + row * ROWB + ((c ^ (row & (CPR - 1) & 7)) * 16));
                    if (ob == 0xFFFFFFFFu || co >= p.OC) continue;
                    T* dst = reinterpret_cast<T*>(base + (size_t)(ob / opix) * pixbytes) + co;
                    if (co + 8 <= p.OC) {
                        *reinterpret_cast<tx8*>(dst) = v;
                    } else {
                        for (int e = 0; e < 8 && co + e < p.OC; ++e) dst[e] = v[e];
                    }
                }
            };
            // ---- the pre-BN tile (kept for the backward pass), while the other blocks arrive ---------------------------------
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        tx4 h;
#pragma unroll
                        for (int r = 0; r < 4; ++r) h[r] = (T)acc[i][j][4 * g + r];
                        const int row = j * 32 + (lane & 31);
                        const int ck = (i * 4 + g) ^ (row & (CPR - 1) & 7);
                        *reinterpret_cast<tx4*>(wt + row * ROWB + ck * 16 + half * 8) = h;
                    }
            store_rows(p.o, opix);
            // ---- phase 2: wait for the grid, fold the bins, normalise + activate the accumulators, store the output ----------
            if (tid == 0) {
                // every block of the launch is resident (the host checked grid <= CUs x occupancy), so the wait is bounded by
                // the slowest block's main loop; the iteration cap only turns a broken assumption into an error flag
                // (ticket[2]) instead of a hung GPU
                int spins = 0;
                while (__hip_atomic_load(p.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nblk_all) {     // (polls bypass the caches)
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > (1 << 23)) {
                        __hip_atomic_store(p.ticket + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // ONE cache invalidation, after the wait
                // the last block to leave puts the ticket back for the next launch (every block has seen it full by then)
                if (__hip_atomic_fetch_add(p.ticket + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblk_all - 1) {
                    __hip_atomic_store(p.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(p.ticket + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __syncthreads();
            // A wait that gave up (ticket[2], set above by the block it happened to) must not pass silently: the
            // statistics may be incomplete, so every block that sees the flag normalises with NaN -- the layer's output, the
            // loss and every gradient of the step turn NaN, which no training loop overlooks.
            const bool broken = __hip_atomic_load(p.ticket + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            float* bnv = reinterpret_cast<float*>(smem + 8 * (WROWS * ROWB));      // [2][BN]: scale, shift (behind the store tiles)
            double* bsum = reinterpret_cast<double*>(bnv + 2 * BN);                  // [2][BN]: sum, sum of squares
            static_assert(8 * WROWS * ROWB + 2 * BN * 4 + 2 * BN * 8 <= LDS_BYTES, "the BN vectors fit behind the store tiles");
            if (tid < 2 * BN) {
                // thread (moment, channel): its CY_STAT_BINS bins requested back to back (independent loads that bypass the
                // caches: the adds were performed at the device's coherence point), then bins in index order, double
                // accumulation -- cy_bn_act_fwd_fused's arithmetic
                const int mom = tid / BN, cl = tid - mom * BN, c = tn * BN + cl;
                float v[CY_STAT_BINS];
#pragma unroll
                for (int b = 0; b < CY_STAT_BINS; ++b)
                    v[b] = c < p.OC ? __hip_atomic_load(p.stats + ((size_t)b * 2 + mom) * p.OC + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
                double t = 0.0;
#pragma unroll
                for (int b = 0; b < CY_STAT_BINS; ++b) t += (double)v[b];
                bsum[tid] = t;
            }
            __syncthreads();
            if (tid < BN) {
                const int c = tn * BN + tid;
                float sc = 0.f, sh = 0.f;
                if (c < p.OC) {
                    const double sm = bsum[tid], sq = bsum[BN + tid];
                    const double cnt = (double)p.M;
                    const double m = sm / cnt;
                    double var = sq / cnt - m * m;
                    if (var < 0.0) var = 0.0;
                    const float is = (float)(1.0 / sqrt(var + (double)p.bn_eps));
                    sc = broken ? __builtin_nanf("") : p.bn_gamma[c] * is;
                    sh = p.bn_beta[c] - (float)m * sc;
                    if (tm == 0) {
                        p.bn_vec[c] = (float)m;
                        p.bn_vec[p.OC + c] = is;
                        p.bn_vec[2 * p.OC + c] = sc;
                        p.bn_vec[3 * p.OC + c] = sh;
                        if (p.bn_rmean) {
                            const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
                            p.bn_rmean[c] = (1.f - p.bn_momentum) * p.bn_rmean[c] + p.bn_momentum * (float)m;
                            p.bn_rvar[c] = (1.f - p.bn_momentum) * p.bn_rvar[c] + p.bn_momentum * (float)unb;
                        }
                        if (p.bn_nbt && c == 0) *p.bn_nbt += 1;
                    }
                }
                bnv[tid] = sc;
                bnv[BN + tid] = sh;
            }
            for (int i = lid * 512 + tid; i < p.bn_zero_n; i += nblk_all * 512) p.bn_zero[i] = 0.f;   // the other table of the pair
            __syncthreads();
            typedef T rx4 __attribute__((ext_vector_type(4)));
            const T* resrow[TJ];
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const unsigned ob = orow[pw + j * 32 + (lane & 31)];
                resrow[j] = (p.res && ob != 0xFFFFFFFFu) ? reinterpret_cast<const T*>(p.res) + (size_t)(ob / opix) * p.ldres : nullptr;
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = cw + i * 32 + 8 * g + 4 * half;
                    const int co = tn * BN + cl;
                    const f32x4 sc4 = *reinterpret_cast<const f32x4*>(bnv + cl), sh4 = *reinterpret_cast<const f32x4*>(bnv + BN + cl);
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
                        float zz[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            // (the accumulator through the storage type first: what the separate pass reads back from `raw`)
                            const float z = (float)(T)acc[i][j][4 * g + r] * sc4[r] + sh4[r];
                            const float zm = mish_f<true>(z), zl = z > 0.f ? z : 0.1f * z;
                            zz[r] = p.act == CY_ACT_MISH ? zm : (p.act == CY_ACT_LEAKY ? zl : z);
                        }
                        if (resrow[j] && co + 3 < p.OC) {
                            const rx4 rv = *reinterpret_cast<const rx4*>(resrow[j] + co);
#pragma unroll
                            for (int r = 0; r < 4; ++r) zz[r] += (float)rv[r];
                        }
                        tx4 h;
#pragma unroll
                        for (int r = 0; r < 4; ++r) h[r] = (T)zz[r];
                        const int row = j * 32 + (lane & 31);
                        const int ck = (i * 4 + g) ^ (row & (CPR - 1) & 7);
                        *reinterpret_cast<tx4*>(wt + row * ROWB + ck * 16 + half * 8) = h;
                    }
                }
            store_rows(p.o2, (unsigned)p.ldo2 * (unsigned)sizeof(T));
            return;
        }
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    tx4 h;
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[r] = (T)acc[i][j][4 * g + r];
                    const int row = j * 32 + (lane & 31);
                    const int ck = (i * 4 + g) ^ (row & (CPR - 1) & 7);
                    *reinterpret_cast<tx4*>(wt + row * ROWB + ck * 16 + half * 8) = h;
                }
        // (wave-private: the wave's own ds_writes are ordered before its ds_reads by lgkmcnt, no barrier needed)
        constexpr int RPI = 64 / CPR;            // pixel rows per store instruction
        // CY_CONV_BNBWD_SUMS: this launch is the last writer of a BN layer's output gradient; its rows pass through here
        // as whole 16-byte chunks, so the BN-backward sums of that layer (sum dz, sum dz * xhat with dz = g act'(z)) are
        // taken on the way out -- one read of the layer's pre-BN tensor instead of a separate pass over (raw, g).
        const bool bnsum = (p.flags & CY_CONV_BNBWD_SUMS) != 0;
        const int cl8 = (lane % CPR) * 8;        // this lane's 8 channels inside the wave tile (the same for every row)
        const bool cok = co_w + cl8 + 8 <= p.OC;
        float bsc[8], bsh[8], s1[8], s2[8];      // s2 collects sum dz * raw; centred and scaled once at the end
        if (bnsum) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ch = min(co_w + cl8 + e, p.OC - 1);
                bsc[e] = p.aff_scale[ch]; bsh[e] = p.aff_shift[ch];
                s1[e] = 0.f; s2[e] = 0.f;
            }
        }
        constexpr int NIT = WROWS / RPI, CHK = NIT <= 8 ? NIT : NIT / 2;
        static_assert(NIT % CHK == 0, "row groups per prefetch chunk");
#pragma unroll
        for (int t0 = 0; t0 < NIT; t0 += CHK) {
            // the pre-BN chunks (and, for a fan-in launch, the gradient already stored) of CHK row groups, requested back
            // to back: one exposed memory latency per chunk instead of one per row group.  (The accumulators are in LDS by
            // now, their registers are free.)
            tx8 rawv[CHK], oldv[CHK];
            unsigned obv[CHK];
#pragma unroll
            for (int t = 0; t < CHK; ++t) {
                obv[t] = orow[pw + (t0 + t) * RPI + lane / CPR];
                const bool ok = obv[t] != 0xFFFFFFFFu && cok;
                if (bnsum) {
                    const T* rp = reinterpret_cast<const T*>(p.res) + (size_t)((ok ? obv[t] : 0u) / ((unsigned)p.ldo * (unsigned)sizeof(T))) * p.ldres + co_w + cl8;
                    rawv[t] = ok ? *reinterpret_cast<const tx8*>(rp) : tx8{};
                }
                if (accum) oldv[t] = ok ? *reinterpret_cast<const tx8*>(reinterpret_cast<const T*>(p.o + obv[t]) + co_w + cl8) : tx8{};
            }
#pragma unroll
            for (int t = 0; t < CHK; ++t) {
                const int row = (t0 + t) * RPI + lane / CPR, c = lane % CPR;
                const unsigned ob = obv[t];
                const int co = co_w + c * 8;
                const tx8 v = *reinterpret_cast<const tx8*>(wt + row * ROWB + ((c ^ (row & (CPR - 1) & 7)) * 16));
                if (ob == 0xFFFFFFFFu || co >= p.OC) continue;
                T* dst = reinterpret_cast<T*>(p.o + ob) + co;
                if (cok) {
                    tx8 o = v;
                    if (accum) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (T)((float)v[e] + (float)oldv[t][e]);
                    }
                    *reinterpret_cast<tx8*>(dst) = o;
                    if (bnsum) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float f = (float)rawv[t][e];
                            const float z = f * bsc[e] + bsh[e];
                            const float dm = mish_grad<true>(z), dl = z > 0.f ? 1.f : 0.1f;
                            const float dz = (float)o[e] * (p.act == CY_ACT_MISH ? dm : (p.act == CY_ACT_LEAKY ? dl : 1.f));
                            s1[e] += dz;
                            s2[e] += dz * f;
                        }
                    }
                } else {
                    for (int e = 0; e < 8 && co + e < p.OC; ++e) dst[e] = (T)((float)v[e] + (accum ? (float)dst[e] : 0.f));
                }
            }
        }
        if (bnsum) {
            // fold the RPI rows a store instruction covers (lanes with equal lane % CPR), publish per (pixel-wave, channel),
            // then one atomic per (channel, moment) of the block -- the table layout of the forward statistics
            float* red = reinterpret_cast<float*>(smem + 8 * (WROWS * ROWB));      // [WM][2][BN], behind the waves' tiles
            static_assert(8 * WROWS * ROWB + WM * 2 * BN * 4 <= LDS_BYTES, "the sums fit behind the store tiles");
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int m = CPR; m < 64; m <<= 1) {
                    s1[e] += __shfl_xor(s1[e], m);
                    s2[e] += __shfl_xor(s2[e], m);
                }
            }
            if (lane < CPR) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int ch = min(co_w + cl8 + e, p.OC - 1);
                    red[(wm * 2 + 0) * BN + cw + cl8 + e] = s1[e];
                    red[(wm * 2 + 1) * BN + cw + cl8 + e] = (s2[e] - p.bn_mean[ch] * s1[e]) * p.bn_invstd[ch];   // sum dz (raw - mean) invstd
                }
            }
            __syncthreads();
            float* srow = p.stats + (size_t)(p.stat_det ? tm : (lid & (CY_STAT_BINS - 1))) * 2 * p.OC;
            for (int c = tid; c < 2 * BN; c += 512) {
                const int mom = c / BN, cl = c - mom * BN, co = tn * BN + cl;
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) t += red[(2 * w + mom) * BN + cl];
                if (co < p.OC) atomicAdd(srow + mom * p.OC + co, t);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const unsigned ob = orow[pw + j * 32 + (lane & 31)];
            if (ob == 0xFFFFFFFFu) continue;
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = co_w + i * 32 + 8 * g + 4 * half;
                    if (co >= p.OC) continue;
                    T* dst = reinterpret_cast<T*>(p.o + ob) + co;
                    if (co + 3 < p.OC) {
                        tx4 h;
                        if (accum) {
                            const tx4 old = *reinterpret_cast<const tx4*>(dst);
#pragma unroll
                            for (int r = 0; r < 4; ++r) h[r] = (T)(acc[i][j][4 * g + r] + (float)old[r]);
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) h[r] = (T)acc[i][j][4 * g + r];
                        }
                        *reinterpret_cast<tx4*>(dst) = h;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (co + r < p.OC) dst[r] = (T)(acc[i][j][4 * g + r] + (accum ? (float)dst[r] : 0.f));
                    }
                }
        }
    }
}

// what a loader wave (no accumulators) owes the block after the main loop: the epilogue's barrier count
__device__ __forceinline__ void pipe_epilogue_loader(const IgemmParams& p) {
    if (p.flags & CY_CONV_STATS) { __syncthreads(); __syncthreads(); }
    if (p.flags & CY_CONV_BNBWD_SUMS) __syncthreads();
    if (p.flags & CY_CONV_BN_FUSED) { __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads(); }
}

// LOADERS = 0: the eight waves stage their own share of every tile.  LOADERS = 4: four extra waves (one per SIMD) do
// nothing but issue the DMA pieces and their scalar / vector address arithmetic, the eight compute waves nothing but
// fragment reads and MFMAs: with two waves per SIMD the ~100 cycles a wave spends issuing each buffer_load ... lds
// (48-64 per K step and CU) are cycles its MFMAs do not issue -- measured, the bare MFMA loop of a 72-step tile takes
// 32 us, 43 us with the DMA issue in the same waves, 51 us with the fragment reads too.
template <typename T, int BM, int BN, int WN, int NST, bool EPI_LDS, int LOADERS>
__global__ void __launch_bounds__(512 + 64 * LOADERS, LOADERS ? 3 : 2) igemm_pipe_kernel(const IgemmParams p) {
    constexpr int WM = 8 / WN;                                // waves over channels x waves over pixels
    constexpr int TI = BN / (32 * WN), TJ = BM / (32 * WM);   // 32 x 32 fragments per wave: channels, pixels
    constexpr int NLD = LOADERS ? LOADERS : 8;                // waves that stage tiles
    constexpr int XP = BM / (8 * NLD), WP = BN / (8 * NLD), NP = XP + WP;   // 1 KB DMA pieces per staging wave per K tile
    static_assert(BM % (8 * NLD) == 0 && BN % (8 * NLD) == 0, "pieces per staging wave");
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int BK = 64;
    static_assert(BM % (32 * WM) == 0 && BM % 64 == 0 && BN % (32 * WN) == 0 && BN % 64 == 0 && NST >= 2 && NST <= 3, "tile / wave layout");
    static_assert(2 * NP <= 48, "vmcnt budget");
    typedef typename Mma32<T>::frag frag;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* orow = reinterpret_cast<unsigned*>(smem + NST * STAGE);   // byte offset of every tile row's output pixel

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_loader = LOADERS && wave >= 8;              // wave-uniform role
    const bool stages = !LOADERS || is_loader;
    const int lw = LOADERS ? (wave - 8) & (NLD - 1) : wave;   // index among the staging waves
    const int wn = (wave & 7) % WN, wm = (wave & 7) / WN;
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, slot = bid >> 3;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    TapSet cls;
    const int tn = lid % p.ntiles, tm = select_class(p, lid / p.ntiles, cls);

    // ---- per-row gather table (one thread per tile row; the staging threads pick their rows up from LDS) -------------
    const int ksign = p.transposed ? -1 : 1;
    const int sh = (p.transposed && p.stride == 2) ? 1 : 0;
    const int span = (p.ks - 1) >> sh;
    const int dmin = p.transposed ? -(span * p.GW + span) * p.ldg * (int)sizeof(T) : 0;
    const int ohw = p.OHc * p.OWc;
    unsigned xoff[XP];
    int ximask[XP];
    {
        uint2* rowinfo = reinterpret_cast<uint2*>(smem);
        if (tid < BM) {
            // the tile covers bm_eff <= BM pixels (the host sizes it so that the grid fills whole rounds of 256 CUs);
            // rows beyond it gather zeros and store nothing
            const int m = tm * p.bm_eff + tid;
            unsigned base = 0u, mask = 0u, obyte = 0xFFFFFFFFu;
            if (tid < p.bm_eff && m < p.M) {
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
                obyte = (unsigned)((n * p.OH + oh) * p.OW + ow) * (unsigned)p.ldo * (unsigned)sizeof(T);
            }
            rowinfo[tid] = make_uint2(base, mask);
            orow[tid] = obyte;
        }
        __syncthreads();
        // staging wave lw owns the pieces lw + NLD * i (8 rows each).  DMA lane l of a piece fills physical chunk l & 7 of
        // row (l >> 3): it fetches logical chunk (l & 7) ^ f(row), f(row) = (row >> 1) & 7 = ((lw & 1) << 2) | (l >> 4)
        const int chunk = (lane & 7) ^ (((lw & 1) << 2) | (lane >> 4));
        const unsigned lane_const = (unsigned)p.x_bias + (unsigned)dmin + (unsigned)(chunk * 16);
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const uint2 ri = rowinfo[(lw + NLD * i) * 8 + (lane >> 3)];
            xoff[i] = ri.x + lane_const;
            ximask[i] = (int)~ri.y;
        }
        __syncthreads();   // stage 0 of the ring overlays rowinfo
    }
    unsigned woff[WP];
    {
        const int chunk = (lane & 7) ^ (((lw & 1) << 2) | (lane >> 4));
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            const int row = tn * BN + (lw + NLD * i) * 8 + (lane >> 3);
            woff[i] = row < p.wrows ? ((unsigned)row * (unsigned)p.K * (unsigned)sizeof(T) + (unsigned)(chunk * 16)) : 0xFFFFFFFFu;
        }
    }
    const auto rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)(p.g - p.x_bias), 0, p.g_bytes + p.x_bias, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

    // wave-uniform K position of the tile whose addresses are being prepared.  Tiles beyond the last one are "dummy":
    // all their pieces carry out-of-range offsets (zero fill into a stage nobody reads any more), so every step issues
    // exactly NP pieces -- one code path, one vmcnt count, no branches in the main loop.
    int l_tap = 0, l_c = 0;
    unsigned x_soff = 0u, w_soff = 0u, ld_oob = 0u;
    int ld_tap = 0;
    unsigned char* ld_dst = smem;
    auto begin_tile = [&](int stage) {      // scalar part of a tile's addresses (SGPRs), then advance the K position
        const bool real = l_tap < cls.ntaps;
        const int tsh = 2 * (real ? l_tap : 0);
        const int kh = (cls.kh_pack >> tsh) & 3, kw = (cls.kw_pack >> tsh) & 3;
        x_soff = (unsigned)((ksign * (((kh >> sh) * p.GW + (kw >> sh)) * p.ldg) + l_c) * (int)sizeof(T) - dmin);
        w_soff = (unsigned)(((kh * p.ks + kw) * p.GC + l_c) * (int)sizeof(T));
        ld_tap = real ? l_tap : 31;         // bit 31 of the inverted tap mask is always set -> every row out of range
        ld_oob = real ? 0u : 0xFFFFFFFFu;
        ld_dst = smem + stage * STAGE + lw * (8 * 128);
        l_c += BK;
        if (l_c >= p.GC) { l_c = 0; ++l_tap; }
    };
    auto issue_piece = [&](int pc) {        // pc: compile-time after unrolling
        if (pc < XP) {
            const unsigned v = xoff[pc] | (unsigned)(-((ximask[pc] >> ld_tap) & 1));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (__attribute__((address_space(3))) void*)(ld_dst + pc * (NLD * 8 * 128)), 16, v,
                                                     x_soff, 0, 0);
        } else {
            const unsigned v = woff[pc - XP] | ld_oob;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rs_w, (__attribute__((address_space(3))) void*)(ld_dst + BM * 128 + (pc - XP) * (NLD * 8 * 128)), 16, v, w_soff, 0, 0);
        }
    };
    auto issue_all = [&]() {
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) issue_piece(pc);
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment reads: lane l reads row (l & 31), logical chunk 2 s + (l >> 5) of K sub-step s -> physical chunk ^ (row >> 1) & 7
    const int qx = ((lane >> 5) ^ ((lane >> 1) & 7)) << 4;
    const int a_row = (BM + wn * (BN / WN) + (lane & 31)) * 128;
    const int b_row = (wm * (BM / WM) + (lane & 31)) * 128;
    frag a[2][TI], b[2][TJ];
    auto load = [&](int set, int stage, int s) {
        const unsigned char* sb = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < TI; ++i) a[set][i] = *reinterpret_cast<const frag*>(sb + a_row + i * (32 * 128) + ((s * 32) ^ qx));
#pragma unroll
        for (int j = 0; j < TJ; ++j) b[set][j] = *reinterpret_cast<const frag*>(sb + b_row + j * (32 * 128) + ((s * 32) ^ qx));
    };
    // the TI * TJ MFMAs of one K sub-step; with FIRST >= 0 the DMA pieces [FIRST, FIRST + COUNT) are issued between them
    // (pinned with sched_barrier fences: hipcc otherwise re-clusters the independent MFMAs)
    auto mma_sub = [&](int set, const int FIRST, const int COUNT) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                acc[i][j] = Mma32<T>::mma(a[set][i], b[set][j], acc[i][j]);
                const int m = i * TJ + j;
#pragma unroll
                for (int k = 0; k < COUNT; ++k)
                    if (FIRST >= 0 && k * (TI * TJ) / COUNT == m) {
                        __builtin_amdgcn_sched_barrier(0);
                        issue_piece(FIRST + k);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
    };
    // pieces of a tile spread over the 4 sub-steps of a K step: sub-step s issues [s * NP / 4, (s + 1) * NP / 4)
    const int nkt = cls.ntaps * (p.GC / BK);

    if constexpr (LOADERS > 0) {
        // Loader / compute split over the same 3-stage ring and the same barrier protocol: at the barrier that opens step kt
        // the loaders have retired their pieces of tile kt (counted vmcnt: tile kt + 1 may still fly) and the compute
        // waves are done with tile kt - 1, whose stage the loaders refill with tile kt + 2 while the compute waves read and
        // multiply tile kt.  Both roles execute exactly one barrier per step.
        static_assert(NST == 3, "the specialised variant runs on the 3-stage ring");
        if (is_loader) {
            begin_tile(0); issue_all();
            begin_tile(1); issue_all();
            begin_tile(2);
            int cs = 0;
            for (int kt = 0; kt < nkt; ++kt) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                issue_all();
                begin_tile(cs);
                cs = cs == 2 ? 0 : cs + 1;
            }
        } else {
            int cs = 0;
            for (int kt = 0; kt < nkt; ++kt) {
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                load(0, cs, 0);
                load(1, cs, 1);
                mma_sub(0, -1, 0);
                load(0, cs, 2);
                mma_sub(1, -1, 0);
                load(1, cs, 3);
                mma_sub(0, -1, 0);
                mma_sub(1, -1, 0);
                cs = cs == 2 ? 0 : cs + 1;
            }
        }
    } else if constexpr (NST == 3) {
        // Top barrier.  Tile kt lives in stage kt % 3.  At the top of step kt this wave has tiles kt and kt + 1 in flight:
        // waiting for vmcnt <= NP retires tile kt (in-order return); the barrier then (a) makes every wave's pieces of
        // tile kt visible and (b) says every wave is done reading stage (kt + 2) % 3, which is refilled during this step.
        // The scalar address part of the tile to issue is prepared at the END of the previous step, so that nothing but
        // the first fragment reads sits between the barrier and the first MFMA.
        begin_tile(0); issue_all();
        begin_tile(1); issue_all();
        begin_tile(2);
        int cs = 0;
        for (int kt = 0; kt < nkt; ++kt) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            load(0, cs, 0);
            load(1, cs, 1);
            mma_sub(0, 0, NP / 4);
            load(0, cs, 2);
            mma_sub(1, NP / 4, NP / 2 - NP / 4);
            load(1, cs, 3);
            mma_sub(0, NP / 2, 3 * NP / 4 - NP / 2);
            mma_sub(1, 3 * NP / 4, NP - 3 * NP / 4);
            begin_tile(cs);              // tile kt + 3 -> the stage just consumed (issued during step kt + 1)
            cs = cs == 2 ? 0 : cs + 1;
        }
    } else {
        begin_tile(0); issue_all();
        begin_tile(1);
        for (int kt = 0; kt < nkt; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int cs = kt & 1;
            load(0, cs, 0);
            load(1, cs, 1);
            mma_sub(0, 0, NP / 4);
            load(0, cs, 2);
            mma_sub(1, NP / 4, NP / 2 - NP / 4);
            load(1, cs, 3);
            mma_sub(0, NP / 2, 3 * NP / 4 - NP / 2);
            mma_sub(1, 3 * NP / 4, NP - 3 * NP / 4);
            begin_tile(cs);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dummy tiles' zero fills must not land on the epilogue's LDS use
    __syncthreads();   // every wave is done with the ring: the epilogue reuses it
    if (is_loader) {   // the loaders hold no results; they only keep the block's barrier count whole
        pipe_epilogue_loader(p);
        return;
    }

    pipe_epilogue<T, BM, BN, WN, EPI_LDS, NST * STAGE>(p, acc, smem, orow, tn, tm, lid);
}

// ---- 3 x 3 / stride 1 / pad 1 with the input staged ONCE per channel chunk ("slab" kernel) ---------------------------------
// The implicit-GEMM kernel above re-fetches the pixel tile for each of the nine taps: per K step (one tap x 64 channels) a
// 256 x 128 tile pulls (256 + 128) x 128 B through the CU's L2 -> LDS path for 4.2 MFLOP, 48 B/clk at the full MFMA rate against
// the 64 B/clk a CU gets -- the round-5 counters have these kernels parked at s_waitcnt / s_barrier half of their wave-cycles
// with LDS only a third busy (profiles/r05_sq_counters.txt).  But the nine taps of a stride-1 3 x 3 convolution read the SAME
// pixels shifted: in the flattened (n, h, w) pixel order tap (kh, kw) of output pixel m is input pixel m + (kh-1) GW + (kw-1).
// So this kernel stages, per 64-channel chunk, ONE slab of bm_eff + 2 GW + 2 consecutive input pixels (the tile plus a halo of one
// image row and one pixel on either side) and serves all nine taps from it by shifting the fragment reads' row index; only the
// weight tile is loaded per K step.  Per chunk a 256 x 128 tile at 38 x 38 moves 334 + 9 x 128 rows instead of 9 x 384: 2.3 x
// fewer bytes and DMA instructions.  Image borders (a shifted pixel that belongs to another image row) are a per-row 9-bit
// mask: a masked lane reads a row of zeros.  Same ring protocol as the loader / compute variant above (3 weight stages, counted
// vmcnt, one raw barrier per K step); the slab of chunk c + 1 is fetched during the first eight steps of chunk c into the
// other of two slab buffers.  K order: chunk-major, tap-minor (the weight matrix's K layout is indexed, not walked).
// Forward and stride-1 dgrad (ksign = -1 mirrors the taps), every epilogue of pipe_epilogue except the two-phase BatchNorm.
template <typename T, int BM, int BN, int WN>
__global__ void __launch_bounds__(768, 3) conv3x3_slab_kernel(const IgemmParams p) {
    constexpr int WM = 8 / WN;
    constexpr int TI = BN / (32 * WN), TJ = BM / (32 * WM);
    constexpr int NLD = 4;                       // loader waves, one per SIMD
    constexpr int WP = BN / (8 * NLD);           // weight pieces (1 KB = 8 rows x 128 B) per loader wave per K step
    constexpr int XPS = 2;                       // slab pieces per loader wave per K step (taps 0..7): room for 64 pieces
    constexpr int WST = BN * 128;                // bytes of one weight stage
    constexpr int W_BYTES = 3 * WST;
    static_assert(BM % (32 * WM) == 0 && BN % (32 * WN) == 0 && BN % (8 * NLD) == 0, "tile / wave layout");
    typedef typename Mma32<T>::frag frag;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int R = p.slab_rows;
    const int XS = R * 128;
    const int ncc = p.GC >> 6;
    const int nsl = ncc > 1 ? 2 : 1;
    // (the tables sit behind whatever is larger: ring + slabs, or what the epilogue re-uses)
    constexpr int EPI_BYTES = BM * BN * 2 + (WM * 8 > 24 ? WM * 8 : 24) * BN;
    const int tabs = W_BYTES + nsl * XS > EPI_BYTES ? W_BYTES + nsl * XS : EPI_BYTES;
    unsigned char* zarea = smem + tabs;                      // 1 KB of zeros: where the dummy pieces land; its first row feeds masked taps
    unsigned* orow = reinterpret_cast<unsigned*>(zarea + 1024);     // [BM] byte offset of every tile row's output pixel
    unsigned* rmask = orow + BM;                                       // [BM] bit t: tap t of the row lies inside its image

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_loader = wave >= 8;
    const int lw = (wave - 8) & (NLD - 1);
    const int wn = (wave & 7) % WN, wm = (wave & 7) / WN;
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, slot = bid >> 3;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int tn = lid % p.ntiles, tm = lid / p.ntiles;
    const int ksign = p.transposed ? -1 : 1;
    const int p0 = tm * p.bm_eff;                 // first pixel of the tile (flattened n, h, w -- input and output alike)

    if (tid < BM) {
        const int m = p0 + tid;
        unsigned mask = 0u, obyte = 0xFFFFFFFFu;
        if (tid < p.bm_eff && m < p.M) {
            const int hw = p.GH * p.GW;
            const int rem = m - (m / hw) * hw;
            const int oh = rem / p.GW, ow = rem - oh * p.GW;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dh = ksign * (t / 3 - 1), dw = ksign * (t % 3 - 1);
                const bool ok = ((unsigned)(oh + dh) < (unsigned)p.GH) & ((unsigned)(ow + dw) < (unsigned)p.GW);
                mask |= (ok ? 1u : 0u) << t;
            }
            obyte = (unsigned)m * (unsigned)p.ldo * (unsigned)sizeof(T);
        }
        orow[tid] = obyte;
        rmask[tid] = mask;
    }
    if (tid < 256) reinterpret_cast<unsigned*>(zarea)[tid] = 0u;
    __syncthreads();

    f32x16 acc[TI][TJ];
    if (is_loader) {
        const auto rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)(p.g - p.x_bias), 0, p.g_bytes + p.x_bias, 0x00020000);
        const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
        // DMA lane l of a piece fills physical chunk l & 7 of row (l >> 3) of its 8 rows: it fetches logical chunk
        // (l & 7) ^ f(row), f(row) = (row >> 1) & 7 = ((piece & 1) << 2) | (l >> 4); this wave's pieces all have piece & 1 = lw & 1
        const int chunk = (lane & 7) ^ (((lw & 1) << 2) | (lane >> 4));
        const unsigned rowbytes = (unsigned)p.ldg * (unsigned)sizeof(T);
        const int npx = R >> 3;
        const int xq = p0 - p.GW - 1 + (lane >> 3);          // flat pixel of this lane's row in piece 0
        const unsigned xlane = p.x_bias + (unsigned)(chunk * 16);
        unsigned woff[WP];
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            const int row = tn * BN + (lw + NLD * i) * 8 + (lane >> 3);
            woff[i] = row < p.wrows ? ((unsigned)row * (unsigned)p.K * (unsigned)sizeof(T) + (unsigned)(chunk * 16)) : 0xFFFFFFFFu;
        }
        // slab piece pc of channel chunk `cc` into slab buffer `sb`; live = false (no such chunk) or pc beyond the slab: a
        // zero fill into the zero area, so that every step issues the same number of pieces (one vmcnt count, no branches)
        auto issue_x = [&](int pc, int sb, int cc, bool live) {
            const int q = xq + 8 * pc;
            const bool in = live && pc < npx;
            const unsigned v = (in && (unsigned)q < (unsigned)p.M) ? xlane + (unsigned)q * rowbytes : 0xFFFFFFFFu;
            unsigned char* dst = in ? smem + W_BYTES + sb * XS + pc * 1024 : zarea;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (__attribute__((address_space(3))) void*)dst, 16, v, (unsigned)(cc * 128), 0, 0);
        };
        // weight tile of K step (cc, tap) into ring stage `st`; cc >= ncc: dummy (zero fill into a stage nobody reads any more)
        auto issue_w = [&](int cc, int tap, int st) {
            const unsigned oob = cc < ncc ? 0u : 0xFFFFFFFFu;
            const unsigned soff = (unsigned)((tap * p.GC + (cc < ncc ? cc : 0) * 64) * (int)sizeof(T));
#pragma unroll
            for (int i = 0; i < WP; ++i) {
                const unsigned v = woff[i] | oob;      // (a named local: with the expression as the argument hipcc's HOST pass drops the kernel's stub)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(smem + st * WST + (lw + NLD * i) * 1024), 16, v,
                                                         soff, 0, 0);
            }
        };
        for (int pc = lw; pc < npx; pc += NLD) issue_x(pc, 0, 0, true);
        issue_w(0, 0, 0);
        issue_w(0, 1, 1);
        for (int cc = 0; cc < ncc; ++cc) {
            const bool live = cc + 1 < ncc;
            const int sb = (cc + 1) & 1;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                // at this barrier the weight tile of step (cc, tap) -- and with tap 0 the whole slab of chunk cc, issued before it --
                // has landed: only what the PREVIOUS step issued may still fly (tap 8 issues no slab pieces)
                if (tap == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WP) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WP + XPS) : "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (tap < 8) {
#pragma unroll
                    for (int k = 0; k < XPS; ++k) issue_x(lw + NLD * (XPS * tap + k), sb, cc + 1, live);
                }
                issue_w(tap + 2 < 9 ? cc : cc + 1, (tap + 2) % 9, (tap + 2) % 3);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the dummy pieces' zero fills must not land on the epilogue's LDS use
        __syncthreads();
        pipe_epilogue_loader(p);
        return;
    }

#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    {
        const int h = lane >> 5;
        const int pw = wm * (BM / WM);
        int crow[TJ];              // slab row of this lane's pixel (centre tap) per fragment
        unsigned tmk[TJ];
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int t = pw + j * 32 + (lane & 31);
            crow[j] = t + p.GW + 1;
            tmk[j] = rmask[t];
        }
        // weights: lane l reads row (l & 31), logical chunk 2 s + (l >> 5) of K sub-step s -> physical chunk ^ (row >> 1) & 7
        const int qa = (h ^ ((lane >> 1) & 7)) << 4;
        const int a_row = (wn * (BN / WN) + (lane & 31)) * 128;
        const unsigned zaddr = (unsigned)(zarea - smem) + (unsigned)(h << 4);
        frag a[2][TI], b[2][TJ];
        unsigned baddr[TJ];
        auto load = [&](int set, int st, int s) {
            const unsigned char* sa = smem + st * WST;
#pragma unroll
            for (int i = 0; i < TI; ++i) a[set][i] = *reinterpret_cast<const frag*>(sa + a_row + i * (32 * 128) + ((s * 32) ^ qa));
#pragma unroll
            for (int j = 0; j < TJ; ++j) b[set][j] = *reinterpret_cast<const frag*>(smem + (baddr[j] ^ (unsigned)(s * 32)));
        };
        auto mma_sub = [&](int set) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = Mma32<T>::mma(a[set][i], b[set][j], acc[i][j]);
        };
        for (int cc = 0; cc < ncc; ++cc) {
            const unsigned xbase = (unsigned)(W_BYTES + (cc & (nsl - 1)) * XS);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dsh = ksign * ((tap / 3 - 1) * p.GW + (tap % 3 - 1));
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const int row = crow[j] + dsh;
                    const unsigned ad = xbase + (unsigned)(row << 7) + (unsigned)((h ^ ((row >> 1) & 7)) << 4);
                    baddr[j] = ((tmk[j] >> tap) & 1u) ? ad : zaddr;
                }
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                load(0, tap % 3, 0);
                load(1, tap % 3, 1);
                mma_sub(0);
                load(0, tap % 3, 2);
                mma_sub(1);
                load(1, tap % 3, 3);
                mma_sub(0);
                mma_sub(1);
            }
        }
    }
    __syncthreads();   // every wave is done with the ring and the slabs: the epilogue reuses them
    pipe_epilogue<T, BM, BN, WN, true, EPI_BYTES, false>(p, acc, smem, orow, tn, tm, lid);
}

template <typename T, int BM, int BN, int WN>
int slab_launch(const IgemmParams& p0, hipStream_t s) {
    IgemmParams p = p0;
    p.mtiles = (p.M + p.bm_eff - 1) / p.bm_eff;
    p.ntiles = (p.OC + BN - 1) / BN;
    p.slab_rows = (p.bm_eff + 2 * p.GW + 2 + 7) & ~7;
    const int nsl = p.GC > 64 ? 2 : 1;
    constexpr int EPI_BYTES = BM * BN * 2 + ((8 / WN) * 8 > 24 ? (8 / WN) * 8 : 24) * BN;
    const int ring = 3 * BN * 128 + nsl * p.slab_rows * 128;
    const int smem = (ring > EPI_BYTES ? ring : EPI_BYTES) + 1024 + BM * 8;
    if (smem > 160 * 1024 || p.slab_rows > 8 * 64) return CY_ERR_UNSUPPORTED;
    static unsigned long long attr_done = 0;      // bit d: set for HIP device d
    if (cy_first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_slab_kernel<T, BM, BN, WN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
    }
    hipLaunchKernelGGL((conv3x3_slab_kernel<T, BM, BN, WN>), dim3(p.mtiles * p.ntiles), dim3(768), smem, s, p);
    CY_LAUNCH_CHECK();
    return 0;
}

// ---- the slab kernel with K split between wave pairs ("slab-K") ----------------------------------------------------------
// Same staging idea as conv3x3_slab_kernel (one halo'd input slab per 64-channel chunk serves the nine taps), other wave layout.
// What the counters and the A/B of the kernels above say (profiles/r06_slab_micro.txt): with the L2 -> LDS traffic cut 2.3 x the
// loop gained 5-9 % -- the limit is inside the CU: a wave of the 8-wave layouts owns a 64 x 64 (or 32 x 96) accumulator block, so
// every 32 x 32 x 16 MFMA needs one ds_read_b128 of its own (4 reads per 4 MFMAs and K sub-step), the first reads after every
// barrier are exposed, and the loader waves take a third of the register file.  Here:
//   * a PAIR of waves shares one accumulator block of twice the size (128 channels x 64 pixels, or 64 x 96) and splits the K step:
//     wave 2u multiplies K sub-steps 0-1 of every 64-channel tile, wave 2u + 1 sub-steps 2-3.  Per wave and K step: 16 MFMAs on
//     12 fragment reads (12 on 10 for the 192-pixel tile) instead of 16 on 16 (12 on 16);
//   * after the K loop the partners exchange halves through LDS (each keeps the channel half the epilogue's wave layout gives it
//     and adds the partner's partial sums): one 16 KB write + read per wave and tile, then pipe_epilogue as it is;
//   * no loader waves: 8 waves x <= 256 registers.  The slab design leaves ~3 DMA pieces per wave and K step;
//   * the ring guarantees tile k + 1 at barrier k (not tile k): a wave reads the first fragments of tile k + 1 while it still
//     multiplies tile k, so no ds_read latency is exposed behind a barrier.  NSTW = 3: one tile in flight behind the two
//     readable ones (the wave drains its own DMA queue at the end of a step -- pieces issued a whole step earlier);
//     NSTW = 4 (where the LDS allows): two in flight, counted vmcnt.
template <typename T, int BM, int BN, int WN, int NSTW>
__global__ void __launch_bounds__(512, 2) conv3x3_slabk_kernel(const IgemmParams p) {
    constexpr int WM = 8 / WN;                   // the EPILOGUE's wave layout: WN channel slices x WM pixel slices
    constexpr int TIE = BN / (32 * WN), TJ = BM / (32 * WM);
    constexpr int TI = 2 * TIE;                  // main loop: a pair (wn = 2 c, 2 c + 1) shares the channel slices 2 c, 2 c + 1
    constexpr int D = NSTW - 2;                  // tiles in flight behind the two readable ones
    constexpr int NWP = BN / 64;                 // weight pieces (1 KB = 8 rows x 128 B) per wave per K step
    constexpr int WST = BN * 128;
    constexpr int W_BYTES = NSTW * WST;
    constexpr int XCH_W = TIE * TJ * 4 * 1024;   // bytes one wave hands to its partner
    static_assert(WN % 2 == 0 && BM % (32 * WM) == 0 && BN % (32 * WN) == 0 && BN % 64 == 0 && NSTW >= 3 && NSTW <= 6, "tile / wave layout");
    typedef typename Mma32<T>::frag frag;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int R = p.slab_rows;
    const int XS = R * 128;
    const int ncc = p.GC >> 6;
    const int nsl = ncc > 1 ? 2 : 1;
    constexpr int EPI_BYTES = BM * BN * 2 + (WM * 8 > 24 ? WM * 8 : 24) * BN;
    constexpr int REUSE = EPI_BYTES > 8 * XCH_W ? EPI_BYTES : 8 * XCH_W;
    const int tabs = W_BYTES + nsl * XS > REUSE ? W_BYTES + nsl * XS : REUSE;
    unsigned char* zarea = smem + tabs;
    unsigned* orow = reinterpret_cast<unsigned*>(zarea + 1024);
    unsigned* rmask = orow + BM;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WN, wm = wave / WN;
    const int kh2 = wn & 1;                      // which half of every K step this wave multiplies; after the loop: which channel half it keeps
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, slot = bid >> 3;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int tn = (CY_ABL & 64) ? lid / p.mtiles : lid % p.ntiles, tm = (CY_ABL & 64) ? lid % p.mtiles : lid / p.ntiles;
    const int ksign = p.transposed ? -1 : 1;
    const int p0 = tm * p.bm_eff;

    if (tid < BM) {
        const int m = p0 + tid;
        unsigned mask = 0u, obyte = 0xFFFFFFFFu;
        if (tid < p.bm_eff && m < p.M) {
            const int hw = p.GH * p.GW;
            const int rem = m - (m / hw) * hw;
            const int oh = rem / p.GW, ow = rem - oh * p.GW;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dh = ksign * (t / 3 - 1), dw = ksign * (t % 3 - 1);
                const bool ok = ((unsigned)(oh + dh) < (unsigned)p.GH) & ((unsigned)(ow + dw) < (unsigned)p.GW);
                mask |= (ok ? 1u : 0u) << t;
            }
            obyte = (unsigned)m * (unsigned)p.ldo * (unsigned)sizeof(T);
        }
        orow[tid] = obyte;
        rmask[tid] = mask;
    }
    if (tid < 256) reinterpret_cast<unsigned*>(zarea)[tid] = 0u;
    __syncthreads();

    // ---- staging: every wave issues NWP weight pieces and (taps 0..7) one slab piece per K step ---------------------------------
    const auto rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)(p.g - p.x_bias), 0, p.g_bytes + p.x_bias, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const int chunk = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));      // this wave's pieces all have piece & 1 = wave & 1
    const unsigned rowbytes = (unsigned)p.ldg * (unsigned)sizeof(T);
    const int npx = R >> 3;
    const int xq = p0 - p.GW - 1 + (lane >> 3);
    const unsigned xlane = p.x_bias + (unsigned)(chunk * 16);
    unsigned woff[NWP];
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
        const int row = tn * BN + (wave + 8 * i) * 8 + (lane >> 3);
        woff[i] = row < p.wrows ? ((unsigned)row * (unsigned)p.K * (unsigned)sizeof(T) + (unsigned)(chunk * 16)) : 0xFFFFFFFFu;
    }
    auto issue_x = [&](int pc, int sb, int cc, bool live) {
        if (CY_ABL & 4) return;
        const int q = xq + 8 * pc;
        const bool in = live && pc < npx;
        const unsigned v = (in && (unsigned)q < (unsigned)p.M) ? xlane + (unsigned)q * rowbytes : 0xFFFFFFFFu;
        unsigned char* dst = in ? smem + W_BYTES + sb * XS + pc * 1024 : zarea;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (__attribute__((address_space(3))) void*)dst, 16, v, (unsigned)(cc * 128), 0, 0);
    };
    auto issue_w = [&](int cc, int tap, int st) {      // weight tile of K step (cc, tap); cc >= ncc: a zero fill nobody reads
        if (CY_ABL & 4) return;
        const unsigned oob = cc < ncc ? 0u : 0xFFFFFFFFu;
        const unsigned soff = (unsigned)((tap * p.GC + (cc < ncc ? cc : 0) * 64) * (int)sizeof(T));
#pragma unroll
        for (int i = 0; i < NWP; ++i) {
            const unsigned v = woff[i] | oob;      // (a named local: see conv3x3_slab_kernel)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(smem + st * WST + (wave + 8 * i) * 1024), 16, v,
                                                     soff, 0, 0);
        }
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int h = lane >> 5;
    const int pw = wm * (BM / WM);
    int crow[TJ];
    unsigned tmk[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int t = pw + j * 32 + (lane & 31);
        crow[j] = t + p.GW + 1;
        tmk[j] = rmask[t];
    }
    const int qa = (h ^ ((lane >> 1) & 7)) << 4;
    const int a_row = ((wn >> 1) * (2 * BN / WN) + (lane & 31)) * 128;
    const unsigned zaddr = (unsigned)(zarea - smem) + (unsigned)(h << 4);
    const int s0 = 2 * kh2;                      // this wave's K sub-steps of every tile: s0, s0 + 1
    frag a[2][TI], b[2][TJ];
    unsigned bcur[TJ], bnext[TJ];
    auto baddr_of = [&](unsigned (&ba)[TJ], int tap, unsigned xbase) {
        const int dsh = ksign * ((tap / 3 - 1) * p.GW + (tap % 3 - 1));
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int row = crow[j] + dsh;
            const unsigned ad = xbase + (unsigned)(row << 7) + (unsigned)((h ^ ((row >> 1) & 7)) << 4);
            ba[j] = ((tmk[j] >> tap) & 1u) ? ad : zaddr;
        }
    };
    auto load = [&](int set, int st, int s, const unsigned (&ba)[TJ]) {
        if (CY_ABL & 2) {
#pragma unroll
            for (int j = 0; j < TJ; ++j) asm volatile("" ::"v"(ba[j]));
            return;
        }
        const unsigned char* sa = smem + st * WST;
#pragma unroll
        for (int i = 0; i < TI; ++i) a[set][i] = *reinterpret_cast<const frag*>(sa + a_row + i * (32 * 128) + ((s * 32) ^ qa));
#pragma unroll
        for (int j = 0; j < TJ; ++j) b[set][j] = *reinterpret_cast<const frag*>(smem + (ba[j] ^ (unsigned)(s * 32)));
    };
    auto vm_wait = [&](int n) {                  // (n folds to a constant after unrolling: one s_waitcnt survives)
        switch (n) {
#define CY_VM(N_) case N_: asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory"); break;
            CY_VM(0) CY_VM(1) CY_VM(2) CY_VM(3) CY_VM(4) CY_VM(5) CY_VM(6) CY_VM(7) CY_VM(8) CY_VM(9) CY_VM(10) CY_VM(11) CY_VM(12)
#undef CY_VM
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };
    auto mma_sub = [&](int set) {
        if (CY_ABL & 1) {
#pragma unroll
            for (int i = 0; i < TI; ++i) asm volatile("" ::"v"(a[set][i]));
#pragma unroll
            for (int j = 0; j < TJ; ++j) asm volatile("" ::"v"(b[set][j]));
            return;
        }
        if (CY_ABL & 128) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[i][j] = Mma32<T>::mma(a[set][i], b[set][j], acc[i][j]);
        if (CY_ABL & 128) __builtin_amdgcn_s_setprio(0);
    };

    // prologue: slab 0, weight tiles 0 .. D; tiles 0 and 1 (and the slab, issued before them) landed at the first barrier
    for (int pc = wave; pc < npx; pc += 8) issue_x(pc, 0, 0, true);
#pragma unroll
    for (int t = 0; t <= D; ++t) issue_w(0, t, t % NSTW);
    vm_wait((D - 1) * NWP);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    baddr_of(bcur, 0, (unsigned)W_BYTES);
    load(0, 0, s0, bcur);

    int st = 0;                                  // ring stage of the current tile (NSTW does not divide 9: a running index)
    for (int cc = 0; cc < ((CY_ABL & 32) ? 0 : ncc); ++cc) {
        const bool live = cc + 1 < ncc;
        const unsigned xb_cur = (unsigned)(W_BYTES + (cc & (nsl - 1)) * XS), xb_nxt = (unsigned)(W_BYTES + ((cc + 1) & (nsl - 1)) * XS);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // barrier k: every wave has waited for its pieces of tile k + 1 and is past its reads of tile k - 1
            if (!(CY_ABL & 8)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // (sched_barrier fences: left alone, hipcc sinks every fragment read to just above its first MFMA and serialises the
            // K step into four read -> wait -> multiply groups)
            auto stage_next = [&]() {
                if (tap < 8) issue_x(wave + 8 * tap, (cc + 1) & 1, cc + 1, live);
                const int t2 = tap + 1 + D;      // tile k + 1 + D -> the stage tile k - 1 has left
                int sn = st + 1 + D;
                sn = sn >= NSTW ? sn - NSTW : sn;
                issue_w(t2 < 9 ? cc : cc + 1, t2 % 9, sn);
            };
            if (CY_ABL & 256) {
                stage_next();
                __builtin_amdgcn_sched_barrier(0);
            }
            load(1, st, s0 + 1, bcur);
            __builtin_amdgcn_sched_barrier(0);
            if (!(CY_ABL & 256)) stage_next();
            baddr_of(bnext, (tap + 1) % 9, tap < 8 ? xb_cur : xb_nxt);
            mma_sub(0);
            __builtin_amdgcn_sched_barrier(0);
            const int s1 = st + 1 >= NSTW ? 0 : st + 1;
            load(0, s1, s0, bnext);
            __builtin_amdgcn_sched_barrier(0);
            mma_sub(1);
#pragma unroll
            for (int j = 0; j < TJ; ++j) bcur[j] = bnext[j];
            st = s1;
            // tile k + 2 landed before the next barrier: only what was issued behind it (the last D - 1 steps' pieces) may fly
            {
                int fly = 0;
#pragma unroll
                for (int d = 0; d + 1 < D; ++d) fly += NWP + (((tap - d + 9) % 9) < 8 ? 1 : 0);
                vm_wait(fly);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // every wave is done with the ring and the slabs
    if (CY_ABL & 16) {
        if (acc[0][0][0] == 123.456f) orow[0] = 1;      // (keeps the loop alive)
        return;
    }

    // ---- partners exchange halves: wave keeps channel slice wn (fragments kh2 * TIE ..), adds the partner's partial sums --------
    f32x16 acc_e[TIE][TJ];
    {
        float* mine = reinterpret_cast<float*>(smem + wave * XCH_W);
        const float* theirs = reinterpret_cast<const float*>(smem + (wave ^ 1) * XCH_W);
        auto give = [&](f32x16 (&blk)[TJ], int fi) {
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = blk[j][4 * q + r];
                    *reinterpret_cast<f32x4*>(mine + (((fi * TJ + j) * 4 + q) * 64 + lane) * 4) = v;
                }
        };
        if (kh2) {
#pragma unroll
            for (int fi = 0; fi < TIE; ++fi) give(acc[fi], fi);
        } else {
#pragma unroll
            for (int fi = 0; fi < TIE; ++fi) give(acc[TIE + fi], fi);
        }
        __syncthreads();
#pragma unroll
        for (int fi = 0; fi < TIE; ++fi)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(theirs + (((fi * TJ + j) * 4 + q) * 64 + lane) * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc_e[fi][j][4 * q + r] = (kh2 ? acc[TIE + fi][j][4 * q + r] : acc[fi][j][4 * q + r]) + v[r];
                }
        __syncthreads();   // the epilogue reuses the exchange area
    }
    pipe_epilogue<T, BM, BN, WN, true, EPI_BYTES, false>(p, acc_e, smem, orow, tn, tm, lid);
}

template <typename T, int BM, int BN, int WN>
int slabk_launch(const IgemmParams& p0, hipStream_t s, int force_nst) {
    IgemmParams p = p0;
    p.mtiles = (p.M + p.bm_eff - 1) / p.bm_eff;
    p.ntiles = (p.OC + BN - 1) / BN;
    p.slab_rows = (p.bm_eff + 2 * p.GW + 2 + 7) & ~7;
    const int nsl = p.GC > 64 ? 2 : 1;
    constexpr int WM = 8 / WN, TIE = BN / (32 * WN), TJ = BM / (32 * WM);
    constexpr int EPI_BYTES = BM * BN * 2 + (WM * 8 > 24 ? WM * 8 : 24) * BN;
    constexpr int XCH = 8 * TIE * TJ * 4 * 1024;
    constexpr int REUSE = EPI_BYTES > XCH ? EPI_BYTES : XCH;
    if (p.slab_rows > 8 * 64) return CY_ERR_UNSUPPORTED;
    auto need = [&](int nst) {
        const int ring = nst * BN * 128 + nsl * p.slab_rows * 128;
        return (ring > REUSE ? ring : REUSE) + 1024 + BM * 8;
    };
    int nst = need(4) <= 160 * 1024 ? 4 : 3;
    if (force_nst == 3 || force_nst == 4) nst = force_nst;      // (5 and 6 stages measured no better: profiles/r06_conv_ablation.txt)
    const int smem = need(nst);
    if (smem > 160 * 1024) return CY_ERR_UNSUPPORTED;
    static unsigned long long attr_done = 0;      // bit d: set for HIP device d
    if (cy_first_use_on_device(attr_done)) {
#define CY_ATTR(N_) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_slabk_kernel<T, BM, BN, WN, N_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        CY_ATTR(3) CY_ATTR(4)
#undef CY_ATTR
    }
    const dim3 grid(p.mtiles * p.ntiles);
    if (nst == 3) hipLaunchKernelGGL((conv3x3_slabk_kernel<T, BM, BN, WN, 3>), grid, dim3(512), smem, s, p);
    else hipLaunchKernelGGL((conv3x3_slabk_kernel<T, BM, BN, WN, 4>), grid, dim3(512), smem, s, p);
    CY_LAUNCH_CHECK();
    return 0;
}

template <typename T, int BM, int BN, int WN, int NST, bool EPI_LDS, int LOADERS = 0>
int pipe_launch(const IgemmParams& p0, hipStream_t s) {
    IgemmParams p = p0;
    p.mtiles = ((p.M + p.bm_eff - 1) / p.bm_eff) * p.ncls;
    p.ntiles = (p.OC + BN - 1) / BN;
    constexpr int smem = NST * (BM + BN) * 128 + BM * 4;
    static_assert(smem <= 160 * 1024, "LDS budget");
    static unsigned long long attr_done = 0;      // bit d: set for HIP device d
    if (cy_first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_pipe_kernel<T, BM, BN, WN, NST, EPI_LDS, LOADERS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
    if (p.flags & CY_CONV_BN_FUSED) {
        // the two-phase epilogue waits for the whole grid: every block must be resident at once
        static int capacity = -1;
        if (capacity < 0) {
            int per_cu = 0, dev = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&igemm_pipe_kernel<T, BM, BN, WN, NST, EPI_LDS, LOADERS>),
                                                             512 + 64 * LOADERS, smem) != hipSuccess ||
                hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
                return CY_ERR_UNSUPPORTED;
            capacity = per_cu * cus;
        }
        if (!EPI_LDS || p.mtiles * p.ntiles > capacity) return CY_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL((igemm_pipe_kernel<T, BM, BN, WN, NST, EPI_LDS, LOADERS>), dim3(p.mtiles * p.ntiles), dim3(512 + 64 * LOADERS), smem, s, p);
    CY_LAUNCH_CHECK();
    return 0;
}

// tile capacities: 384 / 256 / 128 pixels = 4 waves over pixels x 2 over channels, 192 = 2 x 4 (BN = 128 only);
// a 384-pixel tile leaves room for a 2-stage ring only.  variant: 0 eight self-staging waves (3-stage ring, LDS stores),
// 1 direct stores from the MFMA layout, 2 two-stage ring, 3 four loader + eight compute waves.
template <typename T>
int pipe_dispatch(const IgemmParams& p, int cap, int bn, int variant, hipStream_t s) {
    // (the direct-store epilogue of a 384 x 128 tile needs more than 256 VGPRs -- 196 bytes of scratch per lane in round 2 -- and
    // no kernel of the train step may use scratch memory, see build.py: that capacity always stores through LDS)
#define CY_PIPE(BM_, BN_, WN_, NST_)                                                           \
    if (cap == BM_ && bn == BN_) {                                                             \
        if (p.flags & (CY_CONV_BNBWD_SUMS | CY_CONV_BN_FUSED)) {                                               \
            if (variant == 1) return CY_ERR_ARG;   /* the sums / phase 2 live in the LDS-transposed store path */ \
        } else if constexpr (!(BM_ == 384 && BN_ == 128)) {                                                     \
            if (variant == 1 || (p.flags & CY_CONV_ACCUM)) return pipe_launch<T, BM_, BN_, WN_, NST_, false>(p, s); \
        }                                                                                                      \
        if (variant == 2) return pipe_launch<T, BM_, BN_, WN_, 2, true>(p, s);                                \
        if constexpr (NST_ == 3) { if (variant == 3) return pipe_launch<T, BM_, BN_, WN_, 3, true, 4>(p, s); } \
        return pipe_launch<T, BM_, BN_, WN_, NST_, true>(p, s);                                               \
    }
    CY_PIPE(384, 128, 2, 2) CY_PIPE(256, 128, 2, 3) CY_PIPE(192, 128, 4, 3) CY_PIPE(128, 128, 2, 3)
    CY_PIPE(384, 64, 2, 2) CY_PIPE(256, 64, 2, 3) CY_PIPE(128, 64, 2, 3)
#undef CY_PIPE
    return CY_ERR_ARG;
}

}  // namespace

static int64_t g_pipe_launches = 0;
static int g_slab_mode = -1, g_slab_variant = 1, g_slab_nst = 0;
static int g_pipe_mode = -1, g_pipe_cap = 0, g_pipe_bn = 0, g_pipe_variant = 0, g_pipe_bm_eff = 0, g_pipe_bm_eff_slab = 0;

extern "C" int64_t cy_pipe_launches(void) { return g_pipe_launches; }

extern "C" int cy_conv_pipe_config(int mode, int cap, int bn, int variant, int bm_eff) {
    g_pipe_mode = mode; g_pipe_cap = cap; g_pipe_bn = bn; g_pipe_variant = variant; g_pipe_bm_eff = bm_eff;
    return 0;
}

extern "C" int cy_conv_slab_config(int mode, int bm_eff) {
    g_slab_mode = mode & 1; g_pipe_bm_eff_slab = bm_eff;
    g_slab_variant = (mode & 2) ? 0 : 1;          // + 2: the loader / compute variant instead of the K-split wave pairs
    g_slab_nst = (mode & 4) ? 3 : ((mode & 8) ? 4 : 0);      // + 4 / + 8: force the 3- / 4-stage weight ring
    return 0;
}

// Tile policy.  One block per CU, so the grid runs in rounds of 256 tiles and a tile costs ~ (capacity + fixed) whatever
// part of it holds real pixels: for every capacity take the rounds it needs, spread the pixels evenly over the pixel
// tiles that fit those rounds (bm_eff <= capacity), and keep the cheapest (rounds x tile cost).  only_cap != 0 restricts
// the choice to that capacity (the engine's per-layer autotuner names capacities through the CY_CONV_TILE flag field).
static bool pipe_policy(int M, int OC, int only_cap, int& cap, int& bn, int& bm_eff) {
    bn = OC > 64 ? 128 : 64;
    const int ntiles = (OC + bn - 1) / bn;
    static const int caps128[] = {128, 192, 256, 384}, caps64[] = {128, 256, 384};
    const int* caps = bn == 128 ? caps128 : caps64;
    const int ncaps = bn == 128 ? 4 : 3;
    const double fixed = 96.0;      // prologue + epilogue + pipeline fill of one tile, in pixel-equivalents of main loop
    double best = 1e30;
    bool found = false;
    for (int c = 0; c < ncaps; ++c) {
        if (only_cap && caps[c] != only_cap) continue;
        const long tiles = (long)((M + caps[c] - 1) / caps[c]) * ntiles;
        const long r = (tiles + 255) / 256;                 // rounds this capacity needs
        const long mt = 256 * r / ntiles;                   // pixel tiles that fit those rounds
        int eff = (int)((M + mt - 1) / mt);                 // spread the pixels evenly over them
        if (eff < 64) eff = 64;
        if (eff > caps[c]) eff = caps[c];
        const double cost = r * (caps[c] * (caps[c] == 384 ? 1.08 : 1.0) + fixed);   // 384: 2-stage ring only
        if (cost < best) { best = cost; cap = caps[c]; bm_eff = eff; found = true; }
    }
    return found;
}


// The slab kernel's shapes: 3 x 3, stride 1, pad 1 (input and output lattices coincide), 64-channel K chunks, more than 64
// output channels (the 128-channel tile), LDS for ring + slabs.
static int slab_try(const cyk::IgemmParams& p0, int dtype, int hint, hipStream_t s, int* used) {
    if (p0.ks != 3 || p0.stride != 1 || p0.pad != 1 || p0.ncls != 1 || p0.OH != p0.GH || p0.OW != p0.GW) return 0;
    if (p0.flags & (CY_CONV_BN_FUSED | CY_CONV_BIAS_F32OUT)) return 0;
    if ((p0.flags & CY_CONV_BNBWD_SUMS) && (p0.ldres % 8 || ((uintptr_t)p0.res & 15))) return 0;
    if (p0.GC % 64 || !p0.x_bias || p0.OC % 8 || p0.OC <= 64 || p0.ldo % 8) return 0;
    if (((uintptr_t)p0.o & 15) || (p0.res && (p0.ldres % 4 || ((uintptr_t)p0.res & 7)))) return 0;
    if ((size_t)p0.N * p0.OH * p0.OW * p0.ldo * 2 >= 0xFFFFFF00ull) return 0;      // 32-bit output row offsets
    cyk::IgemmParams p = p0;
    int cap = 0, bn = 0, eff = 0;
    const int only = hint == 12 ? 192 : (hint == 13 ? 256 : 0);
    if (only) {
        if (!pipe_policy(p.M, p.OC, only, cap, bn, eff)) return 0;
    } else {
        // policy tile among the two capacities this kernel has
        int c2 = 0, b2 = 0, e2 = 0;
        if (!pipe_policy(p.M, p.OC, 192, cap, bn, eff) || !pipe_policy(p.M, p.OC, 256, c2, b2, e2)) return 0;
        const long r1 = ((long)((p.M + eff - 1) / eff) * ((p.OC + 127) / 128) + 255) / 256, r2 = ((long)((p.M + e2 - 1) / e2) * ((p.OC + 127) / 128) + 255) / 256;
        if (r2 * (256 + 96) < r1 * (192 + 96)) { cap = c2; eff = e2; }
    }
    if (bn != 128) return 0;
    if (g_pipe_bm_eff_slab) eff = g_pipe_bm_eff_slab < cap ? g_pipe_bm_eff_slab : cap;
    p.bm_eff = eff;
    int rc;
    if (g_slab_variant == 0) {
        if (cap == 192) rc = dtype == CY_F16 ? slab_launch<f16, 192, 128, 4>(p, s) : slab_launch<bf16, 192, 128, 4>(p, s);
        else rc = dtype == CY_F16 ? slab_launch<f16, 256, 128, 2>(p, s) : slab_launch<bf16, 256, 128, 2>(p, s);
    } else {
        if (cap == 192) rc = dtype == CY_F16 ? slabk_launch<f16, 192, 128, 4>(p, s, g_slab_nst) : slabk_launch<bf16, 192, 128, 4>(p, s, g_slab_nst);
        else rc = dtype == CY_F16 ? slabk_launch<f16, 256, 128, 2>(p, s, g_slab_nst) : slabk_launch<bf16, 256, 128, 2>(p, s, g_slab_nst);
    }
    if (rc == CY_ERR_UNSUPPORTED) return 0;       // LDS: the slab of a wide image does not fit beside the ring
    if (rc == 0) { *used = 1; ++g_pipe_launches; }
    return rc;
}

// Launches the pipelined kernel when the shape qualifies (*used = 1), else leaves the launch to conv_igemm.hip.
// Which launches take it: the CY_CONV_TILE hint of the call (1: never, 2-5: capacity 128 / 192 / 256 / 384, 6: policy
// tile, 7-9: capacity 128 / 192 / 256 with the loader / compute wave split); without a hint the eval-mode epilogue always does (its LDS-transposed stores are worth 1.3-2x on every shape of
// complex_yolov4.cfg), training launches stay on the 4-wave kernels: measured per layer the two families are within
// +-10 % of each other with the winner depending on how the tiles quantise over 256 CUs, so the engine times both once
// per layer shape and passes the hint (models/engine.py).
int cy_pipe_try(const cyk::IgemmParams& p0, int dtype, hipStream_t s, int* used) {
    *used = 0;
    if (g_pipe_mode < 0) {
        const char* e = getenv("CY_CONV_PIPE");
        g_pipe_mode = e ? atoi(e) : 1;
    }
    int hint = (p0.flags >> CY_CONV_TILE_SHIFT) & 15;
    if (g_slab_mode < 0) {
        const char* e = getenv("CY_CONV_SLAB");
        g_slab_mode = e ? atoi(e) : 1;
    }
    if (hint >= 11 && hint <= 13) {
        // 11-13: the slab kernel (3 x 3, stride 1, pad 1, >= 65 output channels; 11: policy tile, 12 / 13: capacity 192 / 256 pixels);
        // a call it does not take is an ordinary one
        if (g_slab_mode && g_pipe_mode != 0 && (dtype == CY_F16 || dtype == CY_BF16)) {
            const int rc = slab_try(p0, dtype, hint, s, used);
            if (rc || *used) return rc;
        }
        hint = 0;
    }
    if (hint > 9) hint = 0;          // 10: the direct kernels (conv_direct.hip); a call they do not take is an ordinary one here
    if (g_pipe_mode == 0 || hint == 1 || (dtype != CY_F16 && dtype != CY_BF16)) return 0;
    if (g_pipe_mode == 1 && hint == 0 && !(p0.flags & (CY_CONV_AFFINE_ACT | CY_CONV_BNBWD_SUMS | CY_CONV_BN_FUSED))) return 0;
    if ((p0.flags & CY_CONV_BN_FUSED) && (p0.ldo2 % 8 || ((uintptr_t)p0.o2 & 15) || (p0.flags & (CY_CONV_ACCUM | CY_CONV_TRANSPOSED)))) return 0;
    if ((p0.flags & CY_CONV_BNBWD_SUMS) && (p0.ldres % 8 || ((uintptr_t)p0.res & 15))) return 0;
    if (p0.GC % 64 || !p0.x_bias || (p0.flags & CY_CONV_BIAS_F32OUT) || p0.OC % 8 || p0.ldo % 8) return 0;
    if (((uintptr_t)p0.o & 15) || (p0.res && (p0.ldres % 4 || ((uintptr_t)p0.res & 7)))) return 0;
    if ((size_t)p0.N * p0.OH * p0.OW * p0.ldo * 2 >= 0xFFFFFF00ull) return 0;      // 32-bit output row offsets
    cyk::IgemmParams p = p0;
    static const int hint_cap[] = {0, 0, 128, 192, 256, 384, 0, 128, 192, 256};
    int only = (hint >= 2 && hint <= 5) || (hint >= 7 && hint <= 9) ? hint_cap[hint] : 0;
    const bool split = hint >= 7 && hint <= 9;       // four loader waves + eight compute waves
    if (only == 192 && p.OC <= 64) only = 256;
    int cap = 0, bn = 0, eff = 0;
    if (!pipe_policy(p.M * p.ncls, p.OC, only, cap, bn, eff)) return 0;
    if (g_pipe_cap) { cap = g_pipe_cap; bn = g_pipe_bn; eff = g_pipe_bm_eff ? g_pipe_bm_eff : cap; }
    if (eff > cap) return CY_ERR_ARG;
    p.bm_eff = eff;
    const int variant = (split && cap <= 256) ? 3 : g_pipe_variant;
    const int rc = dtype == CY_F16 ? pipe_dispatch<f16>(p, cap, bn, variant, s) : pipe_dispatch<bf16>(p, cap, bn, variant, s);
    if (rc == 0) { *used = 1; ++g_pipe_launches; }
    return rc;
}
