// gconv v2: the gather-convolution implicit GEMM with a VALU-free main loop.
//
// Measured on gfx950 (tools/ubench/mfma_valu.hip): v_mfma_f32_32x32x2_f32
// shares the SIMD's vector datapath -- every VALU instruction, from the same
// or a co-resident wave, delays the MFMA stream by ~4 cycles (pure MFMA loop
// 140 TF/s; +8 v_fma per MFMA: 96 TF/s).  So the K loop here issues, per
// 16-wide K slice and wave, only: MFMAs, ds_read_b128 with immediate offsets,
// `buffer_load_dwordx4 ... lds` (LDS-DMA) with SGPR offsets, one counted
// s_waitcnt and one s_barrier.
//
//   * operands go HBM/L2 -> LDS directly (no VGPR staging, no ds_write) into a
//     ring of NS stages; loads run NS-1 slices ahead of the MFMAs;
//   * the per-lane byte offset (voffset) of every load slot depends only on
//     (row, tap): it is recomputed on tap changes; the channel-chunk offset
//     advances in an SGPR (soffset);
//   * taps that fall outside the frame, zero-inserted positions and rows past
//     M get voffset = 0x80000000: the buffer range check returns zeros;
//   * an LDS-DMA writes lane-linear 1 KiB pieces (16 rows x 64 B); the
//     conflict-free image is obtained by permuting which 16-byte k-quad a lane
//     FETCHES (slot s of row r holds k-quad s ^ ((r>>2)&3)) and applying the
//     same XOR on the read side;
//   * "flat" concat members (NCHW planar or C < 16: the 5-bin voxel grid, the
//     2-channel flow) are few K slices: they run first through a synchronous
//     register path into the same LDS image.
//
// Same GConvParams / epilogue semantics as gconv.hip (v1), which stays as the
// fallback for shapes this kernel does not take (see gconv2_eligible).
#include "conv_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr unsigned OOB = 0x80000000u;

struct KIt {
    int s, tap, c0, coff;
};

}  // namespace

// KSPLIT = 2: 8 waves; wave group g = wave/4 loads and multiplies sub-slice g of
// every K-32 stage (two waves per SIMD even when the grid only offers one
// workgroup per CU), the two accumulator sets are added through LDS at the end.
// TAG only names the instantiation (1: the component GEMMs of winograd.hip, so
// that profiles list them apart from the convolutions proper).
template <int WROWS, int WCOLS, int TM, int TN, int KSUB, int NS, int KSPLIT, int BF16, int TAG = 0>
__global__ __launch_bounds__(CONV_NT *KSPLIT) void gconv2_kernel(const GConvParams P,
                                                                 const int nflat, const int nvec_all)
{
#if defined(__HIP_DEVICE_COMPILE__)   // device-only builtins/types below
    constexpr int BM = WROWS * TM * 32, BN = WCOLS * TN * 32;
    // pieces of 16 rows; the B load tile is padded so that every wave issues
    // the same number of LDS-DMA instructions per stage
    constexpr int PA = BM / 16;
    constexpr int PB0 = (BN + 15) / 16;
    constexpr int PB = PB0 + ((4 - (PA + PB0) % 4) % 4);
    // a stage holds KSUB consecutive 16-channel slices of one tap (K depth
    // 16*KSUB per barrier: halves the per-slice sync cost at 1-2 waves/SIMD)
    static_assert(KSUB % KSPLIT == 0, "wave groups split the sub-slices of a stage");
    constexpr int NT = CONV_NT * KSPLIT;
    constexpr int KPW = KSUB / KSPLIT;            // sub-slices a wave loads and multiplies
    constexpr int LPW = (PA + PB) / 4 * KPW;      // loads per wave per stage
    constexpr int SUB = (PA + PB) * 1024;         // bytes of one 16-wide slice
    constexpr int STAGE = SUB * KSUB;
    constexpr int ROWINFO = NS * STAGE;           // byte offset of rowB/rowY/rowX
    static_assert(WROWS * WCOLS == CONV_NT / kWave, "4 waves");

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    int *rowB = (int *)(smem + ROWINFO), *rowY = rowB + BM, *rowX = rowY + BM;
    long long *rowO = (long long *)(rowX + BM);   // [3][BM] output offsets (conv_epilogue)
    int *rowC = (int *)(rowO + 3 * BM);           // [BM] border class of the output row (bias_cls)

    const int tid = threadIdx.x, lane = tid & 63;
    if (DVSOF_DBG(P) & 512) return;      // probe: launch floor (dispatch + kernarg fetch)
    // wave-uniform values must live in SGPRs: otherwise hipcc wraps every
    // LDS-DMA in a waterfall loop over "possibly divergent" descriptors
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3, wgrp = wave8 >> 2;   // wgrp = sub-slice owned (KSPLIT = 2)
    const int wr = wave / WCOLS, wc = wave % WCOLS;
    // (An XCD-aware block -> tile map for the Winograd component GEMMs -- an XCD
    // gets whole components, so weight and row tiles are fetched into one L2 --
    // measured no change: 34.2 us either way; the probes below show the launch
    // is bound by its fixed costs and by 2.25 workgroups per CU, not by memory.)
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (P.xcd) {
        // Workgroups are dispatched in linear-id order (x fastest), round-robin over the 8
        // XCDs.  Each XCD has its own L2: give XCD i the i-th contiguous eighth of the tiles
        // in (row tile, column tile, phase) order, phase fastest -- the 4 sub-pixel phases and
        // the column tiles of a row tile read the same input rows, now through one L2 and
        // close together in time (bf16 twins: the finest decoder stage fetched its 34 MB of
        // inputs 6 times over, profiles/round3/b_traffic_pmc_bf16s.csv)
        const unsigned gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
        const unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const unsigned t = (L & 7u) * ((gx * gy * gz) >> 3) + (L >> 3);
        bz = (int)(t % gz);
        by = (int)((t / gz) % gy);
        bx = (int)(t / (gz * gy));
    }
    const int m0 = bx * BM, n0 = by * BN;
    const int taps = P.ks * P.ks;
    // exact-tap phases differ 4x in work: the heavy ones are dispatched first
    const int ph = P.ph_exact ? 3 - bz : bz, phy = ph >> 1, phx = ph & 1;
    const int kh = P.ph_exact ? 1 + phy : P.ks, kw = P.ph_exact ? 1 + phx : P.ks;
    const int pad_y = P.pad - phy * P.ph_pad, pad_x = P.pad - phx * P.ph_pad;
    const int nvec = P.ph_exact ? nvec_all / (P.ks * P.ks) * (kh * kw) : nvec_all;
    const float *Wp = P.W + (size_t)ph * P.w_phase_stride;
    const unsigned short *Wp16 = P.W16 ? P.W16 + (size_t)ph * P.w_phase_stride : nullptr;
    const size_t wrow = (size_t)taps * P.Cin_tot;

    for (int r = tid; r < BM; r += NT) {
        const int m = m0 + r;
        int b = 0, y = -(1 << 20), x = -(1 << 20), oy = 0, ox = 0;
        if (m < P.M) {
            if (!P.quad) {
                ox = m % P.Wo;
                const int t = m / P.Wo;
                oy = t % P.Ho;
                b = t / P.Ho;
            } else {
                const int j = m & 3, q = m >> 2, wq = P.Wo >> 1, hq = P.Ho >> 1;
                const int t = q / wq;
                ox = 2 * (q - t * wq) + (j & 1);
                oy = 2 * (t % hq) + (j >> 1);
                b = t / hq;
            }
            y = oy * P.stride - pad_y;
            x = ox * P.stride - pad_x;
        }
        rowB[r] = b;
        rowY[r] = y;
        rowX[r] = x;
        conv_row_offsets(P, rowO, rowC, BM, r, m < P.M, b, oy, ox, phy, phx);
    }
    __syncthreads();
    if (DVSOF_DBG(P) & 1024) return;     // probe: + row tables

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // wave tiles of ONE 32x32 block: a second accumulator for the odd k-quads
    // halves the length of the dependent MFMA chain
    constexpr bool DUAL = (TM * TN == 1) && (BF16 == 0);
    constexpr int EB = BF16 == 3 ? 2 : 4;      // bytes per operand element in HBM and LDS
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;

    // fragment read addresses (bytes inside a stage): row R, k-quad q = 2j + h
    // lives in slot q ^ ((R>>2)&3)
    const int lrow = lane & 31, lh = lane >> 5;
    int a_off[TM][2], b_off[TN][2];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int R = (wr * TM + t) * 32 + lrow;
            a_off[t][j] = R * 64 + (((2 * j + lh) ^ ((R >> 2) & 3)) << 4);
        }
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int R = (wc * TN + t) * 32 + lrow;
            b_off[t][j] = PA * 1024 + R * 64 + (((2 * j + lh) ^ ((R >> 2) & 3)) << 4);
        }

    // Fragments are double buffered in registers: the ds_reads of slice q+1
    // are issued before the MFMAs of slice q, so the LDS latency (and the LDS
    // array time of 4-8 waves reading at once after a barrier) hides behind
    // the matrix pipe instead of stalling it once per slice.
    f32x4 fa[2][2][TM], fb[2][2][TN];
    auto load_frags = [&](auto bufc, const unsigned char *slice) {
        constexpr int buf = decltype(bufc)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int t = 0; t < TM; ++t) fa[buf][j][t] = *(const f32x4 *)(slice + a_off[t][j]);
#pragma unroll
            for (int t = 0; t < TN; ++t) fb[buf][j][t] = *(const f32x4 *)(slice + b_off[t][j]);
        }
    };
    auto mfma_slice = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        if constexpr (BF16 == 3) {
            // bf16 twins: a 16-byte k-quad IS eight bf16 values of K -- one
            // 32x32x16 MFMA per quad, nothing to convert.  A and B read the
            // same quad index, i.e. the same eight k.
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, fa[buf][j][tm]),
                            __builtin_bit_cast(bf16x8, fb[buf][j][tn]), acc[tm][tn], 0, 0, 0);
        } else if constexpr (BF16 != 0) {
            // the lane's two k-quads (8 values) of a 16-wide slice feed ONE
            // 32x32x16 bf16 MFMA; A and B use the same k -> (lane, position) map.
            // BF16 == 2: split operands a = hi + lo (both bf16) and three products
            // hi*hi + hi*lo + lo*hi: the dropped lo*lo term is 2^-16 relative
            bf16x8 ab[TM], bb[TN], al[TM], bl[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const f32x8 v = __builtin_shufflevector(fa[buf][0][t], fa[buf][1][t], 0, 1, 2, 3, 4, 5, 6, 7);
                ab[t] = __builtin_convertvector(v, bf16x8);
                if constexpr (BF16 == 2)
                    al[t] = __builtin_convertvector(v - __builtin_convertvector(ab[t], f32x8), bf16x8);
            }
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const f32x8 v = __builtin_shufflevector(fb[buf][0][t], fb[buf][1][t], 0, 1, 2, 3, 4, 5, 6, 7);
                bb[t] = __builtin_convertvector(v, bf16x8);
                if constexpr (BF16 == 2)
                    bl[t] = __builtin_convertvector(v - __builtin_convertvector(bb[t], f32x8), bf16x8);
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    if constexpr (BF16 == 2) {   // small terms first
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bb[tn], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[tm], bl[tn], acc[tm][tn], 0, 0, 0);
                    }
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[tm], bb[tn], acc[tm][tn], 0, 0, 0);
                }
        } else if constexpr (DUAL) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][0][0][i], fb[buf][0][0][i],
                                                                 acc[0][0], 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][1][0][i], fb[buf][1][0][i], acc2, 0,
                                                            0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                fa[buf][j][tm][i], fb[buf][j][tn][i], acc[tm][tn], 0, 0, 0);
        }
    };
    auto compute1 = [&](const unsigned char *stage) {   // synchronous form (flat members)
        load_frags(std::integral_constant<int, 0>{}, stage);
        mfma_slice(std::integral_constant<int, 0>{});
    };

    // ------------------------------------------------------------------
    // flat concat members first: synchronous register path into stage 0
    // ------------------------------------------------------------------
    if (KSPLIT == 1 && nflat > 0) {
        int s = 0, coff = 0, f0 = 0, done = 0;
        while (!P.src[s].flat) {
            coff += P.src[s].C;
            ++s;
        }
        while (done < nflat) {
            const GSrc &S = P.src[s];
            const int f = f0 + (tid & 15);
            const bool fok = f < taps * S.C;
            const int tap = fok ? f / S.C : 0, c = f - tap * S.C;
            const int ky = tap / P.ks, kx = tap - ky * P.ks;
            const int kk = tid & 15;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < BM / 16; ++i) {
                const int r = (tid >> 4) + 16 * i;
                const int Y = rowY[r] + ky, X = rowX[r] + kx;
                bool ok = fok & ((unsigned)Y < (unsigned)P.Hv) & ((unsigned)X < (unsigned)P.Wv);
                if (P.up == UP_ZERO) ok &= ((Y | X) & 1) == 0;
                float v = 0.f;
                if (ok) {
                    const int ys = P.up ? Y >> 1 : Y, xs = P.up ? X >> 1 : X;
                    v = S.p[(size_t)rowB[r] * S.sb + (size_t)ys * S.sy + (size_t)xs * S.sx +
                            (size_t)c * S.sc];
                }
                if constexpr (BF16 == 3) {   // k = kk lives in quad kk>>3 as bf16; quads 2, 3 are zero
                    unsigned char *row = smem + r * 64;
                    *(unsigned short *)(row + ((((kk >> 3)) ^ ((r >> 2) & 3)) << 4) + (kk & 7) * 2) = bf16_bits(v);
                    *(unsigned short *)(row + ((((kk >> 3) + 2) ^ ((r >> 2) & 3)) << 4) + (kk & 7) * 2) = 0;
                } else
                *(float *)(smem + r * 64 + ((((kk >> 2) ^ ((r >> 2) & 3))) << 4) + (kk & 3) * 4) = v;
            }
#pragma unroll
            for (int i = 0; i < BN / 16; ++i) {
                const int r = (tid >> 4) + 16 * i, n = n0 + r;
                if constexpr (BF16 == 3) {
                    const unsigned short v16 = (fok && n < P.N)
                        ? Wp16[(size_t)n * wrow + (size_t)tap * P.Cin_tot + coff + c] : (unsigned short)0;
                    unsigned char *row = smem + PA * 1024 + r * 64;
                    *(unsigned short *)(row + ((((kk >> 3)) ^ ((r >> 2) & 3)) << 4) + (kk & 7) * 2) = v16;
                    *(unsigned short *)(row + ((((kk >> 3) + 2) ^ ((r >> 2) & 3)) << 4) + (kk & 7) * 2) = 0;
                } else {
                const float v = (fok && n < P.N)
                                    ? Wp[(size_t)n * wrow + (size_t)tap * P.Cin_tot + coff + c] : 0.f;
                *(float *)(smem + PA * 1024 + r * 64 + ((((kk >> 2) ^ ((r >> 2) & 3))) << 4) +
                           (kk & 3) * 4) = v;
                }
            }
            __syncthreads();
            compute1(smem);
            ++done;
            f0 += BK;
            if (f0 >= taps * S.C) {  // next flat member
                f0 = 0;
                coff += S.C;
                ++s;
                while (done < nflat && !P.src[s].flat) {
                    coff += P.src[s].C;
                    ++s;
                }
            }
        }
        __syncthreads();
    }

    // ------------------------------------------------------------------
    // vector members: LDS-DMA ring
    // ------------------------------------------------------------------
    if (nvec > 0) {
        // load slots of this wave: piece p = wave + 4*i, i < LPW
        //   p < PA : A rows 16p .. 16p+15 ; else B rows 16(p-PA) ..
        // lane -> row 16p + (lane>>2), slot lane&3, fetched k-quad slot ^ ((row>>2)&3)
        constexpr int NSLOT = LPW / KPW;    // load slots per wave per 16-wide slice
        int slot_row[NSLOT];
        unsigned slot_kq4[NSLOT];     // byte offset of the fetched k-quad
        unsigned voff[NSLOT];
        int sb_[NSLOT], sy_[NSLOT], sx_[NSLOT];   // row info of A slots
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int p = wave + 4 * i;
            const int r = (p < PA ? 16 * p : 16 * (p - PA)) + (lane >> 2);
            slot_row[i] = r;
            slot_kq4[i] = (unsigned)(((lane & 3) ^ ((r >> 2) & 3)) << 4);
            if (p < PA) {
                sb_[i] = rowB[r];
                sy_[i] = rowY[r];
                sx_[i] = rowX[r];
                voff[i] = OOB;
            } else {
                const int n = n0 + r;
                sb_[i] = sy_[i] = sx_[i] = 0;
                voff[i] = (r < BN && n < P.N) ? (unsigned)(n * wrow * EB) + slot_kq4[i] : OOB;
            }
        }
        const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc(
            BF16 == 3 ? (void *)Wp16 : (void *)Wp, 0, 0x7fffffff, 0x00020000);

        // K-slice iterator, all in SGPRs: member s, tap (ky,kx), chunk c0
        int it_s = 0, it_coff = 0;
        while (P.src[it_s].flat) {
            it_coff += P.src[it_s].C;
            ++it_s;
        }
        // it_c0 / it_C count 4-byte units of a pixel's channel run (f32: channels;
        // bf16 twins: channel pairs); it_coff counts channels of the weight row
        int it_ky = 0, it_kx = 0, it_c0 = 0;
        int it_C = P.src[it_s].C * EB / 4;
        bool new_tap = true;
        long long a_sb = P.src[it_s].sb;
        int a_sy = P.src[it_s].sy, a_sx = P.src[it_s].sx;
        const size_t a_ph = (size_t)ph * P.src_ph_stride;
        auto src_base = [&](int s_) -> void * {
            if constexpr (BF16 == 3) return (void *)(P.src[s_].p16 + a_ph);
            else return (void *)(P.src[s_].p + a_ph);
        };
        __amdgpu_buffer_rsrc_t ares =
            __builtin_amdgcn_make_buffer_rsrc(src_base(it_s), 0, 0x7fffffff, 0x00020000);

        auto issue_impl = [&](auto probec, int stage_idx) {
            constexpr bool PROBE = decltype(probec)::value;   // timing probes compiled in?
            if (new_tap) {  // per-lane offsets of the A slots for this (member, tap)
                new_tap = false;
#pragma unroll
                for (int i = 0; i < NSLOT; ++i) {
                    if (i < PA / 4) {
                        const int Y = sy_[i] + it_ky, X = sx_[i] + it_kx;
                        bool ok = ((unsigned)Y < (unsigned)P.Hv) & ((unsigned)X < (unsigned)P.Wv);
                        if (P.up == UP_ZERO) ok &= ((Y | X) & 1) == 0;
                        const int ys = P.up ? Y >> 1 : Y, xs = P.up ? X >> 1 : X;
                        const unsigned o = (unsigned)(((long long)sb_[i] * a_sb + (long long)ys * a_sy +
                                                       (long long)xs * a_sx) * EB) + slot_kq4[i];
                        voff[i] = ok ? o : OOB;
                    }
                }
            }
#pragma unroll
            for (int sub0 = 0; sub0 < KPW; ++sub0) {
                const int sub = sub0 + wgrp * KPW;
                // dbg 128: every K step re-reads the first chunk (true 64-B-segment access
                // pattern, cache-resident footprint)
                const int a_soff = (PROBE && (DVSOF_DBG(P) & 128)) ? 0 : __builtin_amdgcn_readfirstlane((it_c0 + sub * BK) * 4);
                const int b_soff = (PROBE && (DVSOF_DBG(P) & 128)) ? 0 : __builtin_amdgcn_readfirstlane(
                    ((it_ky * P.ks + it_kx) * P.Cin_tot + it_coff) * EB + (it_c0 + sub * BK) * 4);
                unsigned char *st = smem + stage_idx * STAGE + sub * SUB;
#pragma unroll
                for (int i = 0; i < NSLOT; ++i) {
                    const int p = wave + 4 * i;     // SGPR
                    __attribute__((address_space(3))) void *dst =
                        (__attribute__((address_space(3))) void *)(st + p * 1024);
                    // PA is a multiple of 4 and wave < 4: slot i holds an A piece iff
                    // i < PA / 4 (compile-time: no scalar compare + branch per load)
                    static_assert(PA % 4 == 0, "A pieces per wave");
                    if (PROBE && (DVSOF_DBG(P) & 4)) {   // timing probe: every load hits the same few KiB (L2-resident)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, dst, 16, slot_kq4[i] + (lane >> 2) * 64, 0, 0, 0);
                    } else if (i < PA / 4)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(ares, dst, 16, voff[i], a_soff, 0, 0);
                    else
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, dst, 16, voff[i], b_soff, 0, 0);
                }
            }
            // advance: chunk inner, then tap, then the next vector member
            it_c0 += BK * KSUB;
            if (it_c0 >= it_C) {
                it_c0 = 0;
                new_tap = true;
                if (++it_kx == kw) {
                    it_kx = 0;
                    if (++it_ky == kh) {
                        it_ky = 0;
                        it_coff += P.src[it_s].C;
                        ++it_s;
                        while (it_s < P.nsrc && P.src[it_s].flat) {
                            it_coff += P.src[it_s].C;
                            ++it_s;
                        }
                        if (it_s < P.nsrc) {
                            it_C = P.src[it_s].C * EB / 4;
                            a_sb = P.src[it_s].sb;
                            a_sy = P.src[it_s].sy;
                            a_sx = P.src[it_s].sx;
                            ares = __builtin_amdgcn_make_buffer_rsrc(src_base(it_s), 0, 0x7fffffff,
                                                                     0x00020000);
                        }
                    }
                }
            }
        };

        // prologue: every ring slot in flight, then stage 0's fragments
        static_assert((NS * KPW) % 2 == 0, "static fragment-buffer parity");
#pragma unroll
        for (int u = 0; u < NS; ++u)
            if (u < nvec) issue_impl(std::true_type{}, u);
        if (nvec >= NS) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * LPW) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (DVSOF_DBG(P) & 2048) return;     // probe: + ring prologue (first operands landed)
        load_frags(std::integral_constant<int, 0>{}, smem + wgrp * KPW * SUB);

        // The K loop in two parts: a steady part (every range test of a stage is
        // known to hold: no scalar compares / branches, no probes) and the last
        // stages with the tests.  STEADY is a compile-time flag of the stage body.
        auto stage_body = [&](auto steadyc, auto uc, const int s) {
            constexpr bool STEADY = decltype(steadyc)::value;
            constexpr int u = decltype(uc)::value;
            if (!STEADY && !(s < nvec)) return;
#pragma unroll
            for (int sub = 0; sub < KPW; ++sub) {
                const unsigned char *stage = smem + u * STAGE;
                auto step = [&](auto curc) {
                    constexpr int cur = decltype(curc)::value;
                    if (sub + 1 < KPW) {
                        if (STEADY || !(DVSOF_DBG(P) & 16))
                            load_frags(std::integral_constant<int, cur ^ 1>{},
                                       stage + (sub + 1 + wgrp * KPW) * SUB);
                    } else if (STEADY || s + 1 < nvec) {
                        // stage s+1 has landed when at most the younger stages' loads remain
                        if (STEADY || s + NS - 1 < nvec) {
                            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * LPW) : "memory");
                        } else {
                            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        }
                        // every wave holds its stage-s fragments in registers: slot u is free
                        if (STEADY || !(DVSOF_DBG(P) & 32)) __builtin_amdgcn_s_barrier();
                        if constexpr (STEADY) issue_impl(std::false_type{}, u);
                        else if (s + NS < nvec && !(DVSOF_DBG(P) & 8)) issue_impl(std::true_type{}, u);
                        if (STEADY || !(DVSOF_DBG(P) & 16))
                            load_frags(std::integral_constant<int, cur ^ 1>{},
                                       smem + ((u + 1) % NS) * STAGE + wgrp * KPW * SUB);
                    }
                    mfma_slice(std::integral_constant<int, cur>{});
                };
                if ((u * KPW + sub) & 1) step(std::integral_constant<int, 1>{});
                else step(std::integral_constant<int, 0>{});
            }
        };
        auto ring_turn = [&](auto steadyc, const int s0) {
            stage_body(steadyc, std::integral_constant<int, 0>{}, s0);
            if constexpr (NS > 1) stage_body(steadyc, std::integral_constant<int, 1>{}, s0 + 1);
            if constexpr (NS > 2) stage_body(steadyc, std::integral_constant<int, 2>{}, s0 + 2);
            if constexpr (NS > 3) stage_body(steadyc, std::integral_constant<int, 3>{}, s0 + 3);
            static_assert(NS <= 4, "ring turns are written out for up to 4 stages");
        };
        // steady turns: s0 + NS - 1 + NS < nvec for the last stage of the turn
        const int nsteady = (DVSOF_DBG(P) != 0 || nvec < 2 * NS) ? 0 : (nvec - 2 * NS + 1) / NS * NS;
        int s0 = 0;
        for (; s0 < nsteady; s0 += NS) ring_turn(std::true_type{}, s0);
        for (; s0 < nvec; s0 += NS) ring_turn(std::false_type{}, s0);
    }

    if constexpr (DUAL) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] += acc2[r];
    }
    if (KSPLIT == 2) {   // add the second wave group's accumulators (ring memory is free now)
        __syncthreads();
        float *xch = (float *)smem + (size_t)wave * (TM * TN * 16 * 64);
        if (wgrp == 1) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) xch[((a * TN + b) * 16 + r) * 64 + lane] = acc[a][b][r];
        }
        __syncthreads();
        if (wgrp == 1) return;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] += xch[((a * TN + b) * 16 + r) * 64 + lane];
    }

    if (DVSOF_DBG(P) & 1) return;
    // ---- epilogue (conv_common.h)
    conv_epilogue<TM, TN>(P, acc, rowO, rowC, BM, n0, wr, wc, lane);
#endif  // __HIP_DEVICE_COMPILE__
}

namespace {

template <int WROWS, int WCOLS, int TM, int TN, int KSUB, int NS, int KSPLIT, int BF16, int TAG = 0>
int launch2x(const GConvParams &P, int nflat, int nvec, hipStream_t st)
{
    constexpr int BM = WROWS * TM * 32, BN = WCOLS * TN * 32;
    constexpr int PA = BM / 16, PB0 = (BN + 15) / 16, PB = PB0 + ((4 - (PA + PB0) % 4) % 4);
    constexpr size_t LDS = (size_t)NS * KSUB * (PA + PB) * 1024 + 3 * BM * (sizeof(int) + sizeof(long long)) + BM * sizeof(int);
    static bool attr_set = false;
    if (!attr_set) {
        DVSOF_HIP_TRY(hipFuncSetAttribute(
            (const void *)gconv2_kernel<WROWS, WCOLS, TM, TN, KSUB, NS, KSPLIT, BF16, TAG>,
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        attr_set = true;
    }
    dim3 grid((P.M + BM - 1) / BM, (P.N + BN - 1) / BN, P.nph);
    GConvParams Q = P;
    {   // XCD-aware tile order (see the kernel): DVSOF_GCONV_XCD = 0 off (default: on)
        static const int xe = getenv("DVSOF_GCONV_XCD") ? atoi(getenv("DVSOF_GCONV_XCD")) : -1;
        const unsigned total = grid.x * grid.y * grid.z;
        const bool want = xe != 0;
        // (not the exact-tap phases: they differ 4x in work and are dispatched heavy first;
        // phase-fastest order put heavy ones into the tail: stride-2 data gradients +20-40 %)
        Q.xcd = (want && TAG == 0 && !P.ph_exact && (total & 7u) == 0 && total >= 64) ? 1 : 0;
    }
    hipLaunchKernelGGL((gconv2_kernel<WROWS, WCOLS, TM, TN, KSUB, NS, KSPLIT, BF16, TAG>), grid,
                       dim3(CONV_NT * KSPLIT), LDS, st, Q, nflat, nvec);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

template <int WROWS, int WCOLS, int TM, int TN, int KSUB, int NS, int KSPLIT = 1>
int launch2(const GConvParams &P, int nflat, int nvec, hipStream_t st)
{
    if (P.mfma_bf16 == 3) return launch2x<WROWS, WCOLS, TM, TN, KSUB, NS, KSPLIT, 3>(P, nflat, nvec, st);
    if (P.mfma_bf16 == 2) return launch2x<WROWS, WCOLS, TM, TN, KSUB, NS, KSPLIT, 2>(P, nflat, nvec, st);
    if (P.mfma_bf16 == 1) return launch2x<WROWS, WCOLS, TM, TN, KSUB, NS, KSPLIT, 1>(P, nflat, nvec, st);
    return launch2x<WROWS, WCOLS, TM, TN, KSUB, NS, KSPLIT, 0>(P, nflat, nvec, st);
}

}  // namespace

// v2 takes problems whose vector members have 16-aligned channel counts and
// whose tensors are addressable with 31-bit byte offsets.
bool gconv2_eligible(const GConvParams &P, long long max_src_bytes, long long w_bytes)
{
    bool any_vec = false;
    for (int s = 0; s < P.nsrc; ++s) {
        if (P.src[s].flat) continue;
        any_vec = true;
        if (P.src[s].sc != 1 || (P.src[s].C % BK)) return false;
    }
    if (!any_vec) return false;
    if (max_src_bytes >= 0x7fffffffLL || w_bytes >= 0x7fffffffLL) return false;
    return true;
}

int gconv2_launch(const GConvParams &P0, int tile, hipStream_t st)
{
    GConvParams P = P0;
#ifdef DVSOF_PROBES
    static const int dbg = getenv("DVSOF_GCONV_DBG") ? atoi(getenv("DVSOF_GCONV_DBG")) : 0;
#else
    constexpr int dbg = 0;
#endif
    P.dbg = dbg;
    const int taps = P.ks * P.ks;
    // bf16 twins: every vector member needs its twin, the weights theirs, and a
    // pixel's channel run must be whole 64-byte K slices (32 bf16); else mode 1
    if (P.mfma_bf16 == 3) {
        bool ok = P.W16 != nullptr && P.src_ph_stride == 0 && (P.Cin_tot & 1) == 0;
        for (int s = 0; s < P.nsrc; ++s)
            if (!P.src[s].flat && (!P.src[s].p16 || (P.src[s].C % (2 * BK)))) ok = false;
        if (!ok) P.mfma_bf16 = 1;
    }
    const int unit = P.mfma_bf16 == 3 ? 2 : 1;     // channels per 4-byte unit of a K slice
    // K depth per barrier: 32 when every vector member allows it (small tiles
    // run 1-2 waves per SIMD, where the per-slice sync cost is exposed)
    static const bool k16 = getenv("DVSOF_GCONV_K16") != nullptr;
    bool k32 = !k16 && (tile == 2 || tile == 3);
    for (int s = 0; s < P.nsrc; ++s)
        if (!P.src[s].flat && ((P.src[s].C / unit) % (2 * BK))) k32 = false;
    {   // larger stages cost occupancy: only when the grid is <= 2 workgroups per CU anyway
        const long long bm = 128 >> (tile == 3), bn = 64;
        const long long blocks = ((P.M + bm - 1) / bm) * ((P.N + bn - 1) / bn) * P.nph;
        static const long long k32_blocks = getenv("DVSOF_GCONV_K32_BLOCKS") ? atoll(getenv("DVSOF_GCONV_K32_BLOCKS")) : 2 * 256;
        if (blocks > k32_blocks) k32 = false;
    }
    int ksub = k32 ? 2 : 1;
    // one workgroup per CU (8-wave form): K depth 64 per barrier -- both waves of
    // a SIMD reach the barrier together, so the sync/issue bubble is paid per stage
    bool k64 = false;
    if (tile == 3 && k32 && !getenv("DVSOF_GCONV_NO_K64")) {
        const long long blocks = ((P.M + 63) / 64) * ((P.N + 63) / 64) * P.nph;
        k64 = blocks <= 256;
        for (int s = 0; s < P.nsrc; ++s)
            if (P.src[s].flat || ((P.src[s].C / unit) % (4 * BK))) k64 = false;
        if (k64) ksub = 4;
    }
    int nflat = 0, nvec = 0;
    for (int s = 0; s < P.nsrc; ++s) {
        if (P.src[s].flat) nflat += (taps * P.src[s].C + BK - 1) / BK;
        else nvec += taps * (P.src[s].C / unit / (BK * ksub));
    }
    if (dbg & 2) nvec = nvec > 1 ? 1 : nvec;
    if (nflat > 0 || getenv("DVSOF_NO_PH_EXACT")) P.ph_exact = 0;
    // Winograd component GEMMs (winograd.hip): 64 x 64, K depth 16 (measured best of the
    // tiles / depths, tools/wino_sweep.sh), under their own kernel name
    if (P.src_ph_stride != 0 && tile == 3 && !k32 && nflat == 0) {
        if (P.mfma_bf16 == 2) return launch2x<2, 2, 1, 1, 1, 4, 1, 2, 1>(P, nflat, nvec, st);
        if (P.mfma_bf16 == 0) return launch2x<2, 2, 1, 1, 1, 4, 1, 0, 1>(P, nflat, nvec, st);
    }
    switch (tile) {
    case 1: return launch2<2, 2, 2, 2, 1, 4>(P, nflat, nvec, st);  // 128 x 128
    case 2: return k32 ? launch2<2, 2, 2, 1, 2, 3>(P, nflat, nvec, st)
                       : launch2<2, 2, 2, 1, 1, 4>(P, nflat, nvec, st);  // 128 x 64
    case 3: {
        // few workgroups per CU: 8-wave form (two waves per SIMD from one workgroup)
        static const bool no8 = getenv("DVSOF_GCONV_NO_KSPLIT") != nullptr;
        const long long blocks = ((P.M + 63) / 64) * ((P.N + 63) / 64) * P.nph;
        // ring depth 2 (64 KiB): a deeper ring is no faster stand-alone and its LDS
        // footprint keeps the second stream's workgroups off the CU
        if (k64 && !no8 && P.mfma_bf16 == 3) {
            // bf16 twins: 4 matrix instructions per wave and stage -- the loop is bound
            // by the latency of the LDS-DMA stream, i.e. by the bytes in flight
            // ((NS - 1) stages of 32 KiB), not by the matrix pipe as in f32
            static const int ns3 = getenv("DVSOF_GCONV_K64_NS") ? atoi(getenv("DVSOF_GCONV_K64_NS")) : 4;
            if (ns3 == 4) return launch2x<2, 2, 1, 1, 4, 4, 2, 3>(P, nflat, nvec, st);
            if (ns3 == 3) return launch2x<2, 2, 1, 1, 4, 3, 2, 3>(P, nflat, nvec, st);
        }
        if (k64 && !no8) return launch2<2, 2, 1, 1, 4, 2, 2>(P, nflat, nvec, st);
        if (k64) { /* 4-wave fallback keeps K32 counting */ return DVSOF_EINVAL; }
        // (<= 2 workgroups per CU: measured +3 % on the 512-workgroup decoder layers over the 4-wave form)
        static const long long ks_blocks = getenv("DVSOF_GCONV_KSPLIT_BLOCKS") ? atoll(getenv("DVSOF_GCONV_KSPLIT_BLOCKS")) : 512;
        if (k32 && !no8 && nflat == 0 && blocks <= ks_blocks)
            return launch2<2, 2, 1, 1, 2, 4, 2>(P, nflat, nvec, st);
        return k32 ? launch2<2, 2, 1, 1, 2, 4>(P, nflat, nvec, st)
                   : launch2<2, 2, 1, 1, 1, 4>(P, nflat, nvec, st);  // 64 x 64
    }
    case 4: return launch2<4, 1, 2, 1, 1, 4>(P, nflat, nvec, st);  // 256 x 32
    case 5: return launch2<4, 1, 1, 1, 1, 4>(P, nflat, nvec, st);  // 128 x 32
    default: return DVSOF_EINVAL;
    }
}
