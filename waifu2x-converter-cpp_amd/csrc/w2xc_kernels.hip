// w2xc_kernels.hip -- gfx950 (MI355X, CDNA4) kernels for the waifu2x 3x3-conv + bias + LeakyReLU
// layer, i.e. the body of Model::filterWorker (/root/reference/src/modelHandler.cpp:117-159).
//
// Four kernels, one launch per layer per band (see w2xc_kernels.h for the addressing contract):
//
//   conv3x3_direct      any (cin, cout), any strides.  VALU, one thread per output pixel and 8
//                       output planes; sums in the REFERENCE'S order with unfused mul/add, so it
//                       is bit-exact against the CPU oracle.  Fallback + on-GPU cross-check.
//   conv3x3_mfma2       cin, cout in {32,64,128}.  NHWC fp32 activations; persistent workgroups (one per CU); the
//                       10 x 34 pixel halo tile of a 32-channel slice and the weights of one (slice, tap) stage reach
//                       LDS by LDS-DMA (XOR-swizzled: conflict-free ds_read_b128), the 3x3xCinxCout contraction is an
//                       implicit GEMM on v_mfma_f32_32x32x2_f32 (M = 32 pixels of one row, N = 32 output planes,
//                       K = (tap, cin)), bias + LeakyReLU fused into the epilogue, NHWC stores in 128-byte runs.
//                       (The first-generation kernel it replaced -- register-staged tiles, weights from L2 -- was
//                       removed in round 2; its history is in DESIGN.md 3.)
//   conv3x3_first       cin <= 3 (layer 1): planar input, clamp-to-edge folded into the LDS fill
//                       (this is cv::copyMakeBorder, convertRoutine.cpp:35), K = 9*cin on the same
//                       32x32x2 MFMA, NHWC output.  HBM-write bound.
//   conv3x3_last        cout <= 3 (layer 7): "taps as N": G[q][tap,o] = sum_c in[q][c]*W[o][c][tap]
//                       for every haloed pixel q on v_mfma_f32_16x16x4_f32 straight from global
//                       memory, then out[p][o] = sum_tap G[p+tap][tap,o] through LDS; planar
//                       output written at its final place (the crop/stitch of
//                       convertRoutine.cpp:40-46,143-161).  HBM-read bound.
//
// fp32 MFMA on gfx950 is an exact k-ordered fmaf chain at the f32 vector rate (157.3 TF peak).
#include "w2xc_kernels.h"
#include "w2xc_device.h"

#include <stdlib.h>
#include <string.h>

#include <atomic>


// ------------------------------------------------------------------------------------------------
// conv3x3_direct: reference-ordered VALU kernel (bit-exact vs the oracle).
//   weights packed [cin][9][cout8-padded]: for one (i, tap) the 8 output planes of a group are
//   contiguous and wave-uniform -> scalar loads.
// ------------------------------------------------------------------------------------------------
#define DIRECT_CG 8
__global__ void __launch_bounds__(256) conv3x3_direct(W2xcConvDesc d, int cout_pad)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int og = blockIdx.z * DIRECT_CG;
    if (x >= d.out_w || y >= d.out_h) return;
    long long roff[3], coff[3];
#pragma unroll
    for (int r = 0; r < 3; r++) roff[r] = (long long)(clampi(y + r + d.off_y, 0, d.in_h - 1) >> d.in_shift) * d.in_rs;
#pragma unroll
    for (int c = 0; c < 3; c++) coff[c] = (long long)(clampi(x + c + d.off_x, 0, d.in_w - 1) >> d.in_shift) * d.in_ps;
    float acc[DIRECT_CG];
#pragma unroll
    for (int k = 0; k < DIRECT_CG; k++) acc[k] = 0.0f;                                  // :131-132
    for (int i = 0; i < d.cin; i++) {                                                   // :134
        const float *ip = d.in + (long long)i * d.in_cs;
        float v[9];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) v[r * 3 + c] = ip[roff[r] + coff[c]];
        const float *wp = d.wpk + ((long long)i * 9) * cout_pad + og;
        float t[DIRECT_CG];
#pragma unroll
        for (int k = 0; k < DIRECT_CG; k++) t[k] = 0.0f;                                // filter2D delta = 0
#pragma unroll
        for (int tap = 0; tap < 9; tap++)                                               // row-major taps
#pragma unroll
            for (int k = 0; k < DIRECT_CG; k++)
                t[k] = __fadd_rn(t[k], __fmul_rn(wp[tap * cout_pad + k], v[tap]));     // mul, then add
#pragma unroll
        for (int k = 0; k < DIRECT_CG; k++) acc[k] = __fadd_rn(acc[k], t[k]);           // cv::add :144
    }
    float *op = d.out + (long long)y * d.out_rs + (long long)x * d.out_ps;
#pragma unroll
    for (int k = 0; k < DIRECT_CG; k++) {
        if (og + k < d.cout) {
            const float v = __fadd_rn(acc[k], d.bias[og + k]);                           // :147
            const float pos = v > 0.0f ? v : 0.0f, neg = v < 0.0f ? v : 0.0f;            // :150-151
            op[(long long)(og + k) * d.out_cs] = __fadd_rn(__fmul_rn(neg, 0.1f), pos);   // :152
        }
    }
}

// ------------------------------------------------------------------------------------------------
// conv3x3_mfma2: second-generation fp32 MFMA implicit GEMM -- every MFMA operand comes from LDS and
// every byte gets there by LDS-DMA (global_load_lds_dwordx4), so the matrix-core instruction stream
// only ever waits on lgkmcnt; the in-order vmcnt queue holds nothing but bulk transfers issued two
// or more stages ahead and the epilogue stores.
//
//   Persistent workgroup (1 per CU, 4 waves, one per SIMD, up to 512 registers each), tile = 8 rows
//   x 32 pixels x COUT planes; wave (wm, wn) owns MB row-blocks x NB plane-blocks (MB*WM = 8).
//   Stage = (32-channel slice, tap): 4 k-groups of 8 channels = 16*MB*NB MFMAs per wave.
//   LDS:  A[2] x 44 KiB  halo tile (10 x 34 pixels x 32 channels) of the current / next slice,
//                        UNPADDED 128 B per pixel, 16-byte chunk q of pixel p stored at chunk
//                        position q ^ ((p>>1)&7)  (DMA writes are lane-linear, so the swizzle is
//                        applied to the per-lane SOURCE address; ds_read_b128 of 32 consecutive
//                        pixels is then conflict-free in all four 16-lane groups)
//         B[4] x 4*NBT KiB  the weights of one stage in MFMA fragment order (a straight copy of
//                        wpk[tap][4 k-groups][NBT][64 lanes][4]); ring of 4 stages.
//   Schedule at stage t:  issue DMA B(t+3) and up to 2 pieces of A(next slice | next tile);
//   at its end wait (counted vmcnt) for this wave's share of B(t+2) [and A(next) when t = 7],
//   raw s_barrier.  B(t+1) was already complete at the previous barrier, so the first fragments
//   of stage t+1 are read BEFORE the barrier and the MFMA stream runs across it.
// ------------------------------------------------------------------------------------------------

template <int CIN, int COUT, int MB, int NB, int WM, int WN, int EPI = 1>
__global__ void __launch_bounds__(WM *WN * 64, WM *WN / 4) conv3x3_mfma2(W2xcConvDesc d, int tiles_x, int ntiles)
{
    constexpr int NST = MB * NB * 16;                // stores per wave in an interior-tile epilogue
    constexpr int ROWS = 8, HW = 34, HH = ROWS + 2, NPIX = HH * HW;
    constexpr int NSL = CIN / 32, NBT = COUT / 32;
    constexpr int NW = WM * WN;                      // 4 waves (one per SIMD) or 8 (two per SIMD)
    constexpr int APW = (43 + NW) / NW;              // A pieces (1 KiB) per wave per slice: NW*APW >= 42.5
    constexpr unsigned A_BYTES = NW * APW * 1024;    // 45056 (4 waves) / 49152 (8 waves)
    constexpr int BPW = 4 * NBT / NW;                // B pieces per wave per stage (4*NBT in total)
    constexpr unsigned B_BYTES = 4 * NBT * 1024;
    constexpr unsigned B_BASE = 2 * A_BYTES;
    static_assert((NW == 4 || NW == 8) && MB * WM == ROWS && NB * WN == NBT && BPW * NW == 4 * NBT, "tile shape");
    static_assert(CIN % 32 == 0 && COUT % 32 == 0, "planes");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    const char *ldsb = reinterpret_cast<const char *>(lds);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int nb0 = wn * NB;
    const int li = lane & 31, kk = lane >> 5;

    // persistent schedule: XCD x (= blockIdx % 8) walks its own contiguous chunk of the tile list
    const int xcd = blockIdx.x & 7, per = gridDim.x >> 3;
    const int cq = ntiles >> 3, cr = ntiles & 7;
    const int chunk_begin = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
    const int chunk_end = chunk_begin + cq + (xcd < cr ? 1 : 0);
    int tile = chunk_begin + (blockIdx.x >> 3);
    if (tile >= chunk_end) return;

    float bv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; nb++) bv[nb] = d.bias[(nb0 + nb) * 32 + li];

    // ---- per-lane DMA source offsets of the A halo tile (16-byte units), piece jj of this wave ----
    const f32x4 *in4 = reinterpret_cast<const f32x4 *>(d.in);
    // goff[jj] = lofs[jj] + (tile origin) for tiles whose halo needs no clamping (all but the border tiles):
    // lofs is tile-independent, so the per-tile bookkeeping is APW adds instead of ~20 VALU per piece.
    unsigned goff[APW], lofs[APW];
#pragma unroll
    for (int jj = 0; jj < APW; jj++) {
        const int s = (jj * NW + wave) * 64 + lane;
        int p = s >> 3;
        p = p < NPIX ? p : NPIX - 1;
        const int py = p / HW, px = p - py * HW;
        lofs[jj] = (unsigned)(((long long)py * d.in_rs + (long long)px * CIN) >> 2) + ((s & 7) ^ ((p >> 1) & 7));
    }
    auto tile_offsets = [&](int t) {
        const int ty_ = t / tiles_x, tx_ = t - ty_ * tiles_x;
        const int y0 = ty_ * ROWS + d.off_y, x0 = tx_ * 32 + d.off_x;
        if (y0 >= 0 && y0 + HH <= d.in_h && x0 >= 0 && x0 + HW <= d.in_w) {   // wave-uniform
            const unsigned base = (unsigned)(((long long)y0 * d.in_rs + (long long)x0 * CIN) >> 2);
#pragma unroll
            for (int jj = 0; jj < APW; jj++) goff[jj] = lofs[jj] + base;
            return;
        }
#pragma unroll
        for (int jj = 0; jj < APW; jj++) {
            const int s = (jj * NW + wave) * 64 + lane;           // 16-byte slot in the A buffer
            int p = s >> 3;
            p = p < NPIX ? p : NPIX - 1;                           // slots past the tile re-read its last pixel
            const int q = (s & 7) ^ ((p >> 1) & 7);                // chunk stored at this position
            const int py = p / HW, px = p - py * HW;
            const int gy = clampi(y0 + py, 0, d.in_h - 1);
            const int gx = clampi(x0 + px, 0, d.in_w - 1);
            goff[jj] = (unsigned)(((long long)gy * d.in_rs + (long long)gx * CIN) >> 2) + q;
        }
    };
    const unsigned b_voff = (unsigned)((wave * BPW) * 64 + lane) * 16u;                      // this wave's B pieces
    auto dma_b = [&](int sl_, int tap_, unsigned buf, int jb) {   // piece jb of stage (sl_, tap_) -> ring slot buf
        const char *sbase = reinterpret_cast<const char *>(d.wpk) + (size_t)(tap_ * (CIN / 8) + sl_ * 4) * NBT * 1024;
        const unsigned dst = lds0 + B_BASE + buf * B_BYTES + (unsigned)(wave * BPW) * 1024u;
        switch (jb) {
        case 0: lds_dma16_s<0>(sbase, b_voff, dst); break;
        // the instruction's immediate offset is applied to BOTH the global and the LDS address
        case 1: lds_dma16_s<1024>(sbase, b_voff, dst); break;
        case 2: lds_dma16_s<2048>(sbase, b_voff, dst); break;
        default: lds_dma16_s<3072>(sbase, b_voff, dst); break;
        }
    };
    auto dma_a = [&](unsigned add, unsigned abuf, int jj) {
        lds_dma16(in4 + goff[jj] + add, lds0 + abuf * A_BYTES + (unsigned)(jj * NW + wave) * 1024u);
    };

    // ---- fragment addressing ----
    // A: lane (li, kk) reads chunk q = 2*c8 + kk of pixel p = (wm*MB + row)*34 + li + tx, row = mb + ty:
    //    byte p*128 + ((q ^ ((p>>1)&7)) << 4) = a0[row][tx] ^ (c8 << 5)  with a0 the c8 = 0 address
    //    (bits 0..6 of p*128 are zero and 2*c8 only touches chunk bits 1..2).  a0 is tile-independent.
    unsigned a0[MB + 2][3];
#pragma unroll
    for (int row = 0; row < MB + 2; row++)
#pragma unroll
        for (int tx = 0; tx < 3; tx++) {
            const int p = (wm * MB + row) * HW + li + tx;
            a0[row][tx] = (unsigned)(p * 128 + ((((p >> 1) & 7) ^ kk) << 4));
        }
    auto a_addr = [&](unsigned abuf, int mb, int tap, int c8) -> const f32x4 * {
        return reinterpret_cast<const f32x4 *>(ldsb + ((a0[mb + tap / 3][tap % 3] ^ (unsigned)(c8 << 5)) + abuf * A_BYTES));
    };
    auto b_addr = [&](unsigned buf, int c8, int nb) -> const f32x4 * {
        return reinterpret_cast<const f32x4 *>(ldsb + B_BASE + buf * B_BYTES + ((c8 * NBT + nb0 + nb) * 64 + lane) * 16);
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; mb++)
#pragma unroll
        for (int nb = 0; nb < NB; nb++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mb][nb][r] = 0.0f;

    // ---- prologue: A(slice 0) and B stages 0..2 of the first tile ----
    tile_offsets(tile);
#pragma unroll
    for (int jj = 0; jj < APW; jj++) dma_a(0, 0, jj);
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int jb = 0; jb < BPW; jb++) dma_b((t / 9) % NSL, t % 9, t & 3, jb);
    W2XC_WAIT_VMCNT(0);
    __builtin_amdgcn_s_barrier();

    unsigned gs = 0;      // global stage counter (only gs & 3 matters): ring slot of the current stage
    unsigned abuf = 0;    // A buffer of the current slice
    int sl = 0;
    bool epi_stores = false;   // an interior-tile epilogue (NST stores) directly precedes the current stage
    f32x4 a_cur[MB], b_cur[NB];
#pragma unroll
    for (int mb = 0; mb < MB; mb++) a_cur[mb] = *a_addr(0, mb, 0, 0);
#pragma unroll
    for (int nb = 0; nb < NB; nb++) b_cur[nb] = *b_addr(0, 0, nb);

    for (;;) {
        // what the A pieces issued during this slice fetch: the next slice, or the next tile's first
        const bool last_slice = (sl == NSL - 1);
        unsigned a_add = (unsigned)(sl + 1) * 8;
        if (last_slice) {
            tile_offsets(tile + per < chunk_end ? tile + per : tile);
            a_add = 0;
        }
        const int sl_next = last_slice ? 0 : sl + 1;

#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            // stage (sl, tap) in ring slot gs&3; DMA targets: B of stage +3, A pieces 2 per stage on taps 0..5
            const int tap3 = (tap + 3) % 9;
            const int sl3 = (tap + 3 < 9) ? sl : sl_next;
            const unsigned buf = gs & 3u, buf3 = (gs + 3u) & 3u, buf1 = (gs + 1u) & 3u;
            // A pieces of the next slice are issued on taps 0..5: KA[t] = ceil-spread of APW over 6 stages
            constexpr int KA[9] = {(APW + 5) / 6, (APW + 4) / 6, (APW + 3) / 6, (APW + 2) / 6, (APW + 1) / 6, APW / 6, 0, 0, 0};
            int ja0 = 0;               // first A piece of this stage
#pragma unroll
            for (int t = 0; t < 9; t++) ja0 += t < tap ? KA[t] : 0;
#pragma unroll
            for (int c8 = 0; c8 < 4; c8++) {
                // One step = M MFMAs.  Every other instruction is pinned into an MFMA shadow:
                //   the MB + NB fragment reads of the NEXT step (the last step of a stage reads the next
                //   stage's first fragments, which the previous barrier already guaranteed), spaced evenly;
                //   the DMAs: step 0: [A piece, A piece,] B piece 0; steps 1..3: B piece c8.
                constexpr int M = 4 * MB * NB, R = MB + NB;
                const bool wrap = (c8 == 3);
                const int tap_n = wrap ? (tap + 1) % 9 : tap, c8_n = wrap ? 0 : c8 + 1;
                const unsigned abuf_n = (wrap && tap == 8) ? (abuf ^ 1u) : abuf;
                const unsigned bbuf_n = wrap ? buf1 : buf;
                f32x4 a_nxt[MB], b_nxt[NB];
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int mb = 0; mb < MB; mb++)
#pragma unroll
                        for (int nb = 0; nb < NB; nb++) {
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[mb][j], b_cur[nb][j], acc[mb][nb], 0, 0, 0);
                            const int m = (j * MB + mb) * NB + nb;        // MFMA index in the step
                            if (c8 == 0 && (m == 1 || m == 3) && (m >> 1) < KA[tap]) {
                                __builtin_amdgcn_sched_barrier(0);
                                dma_a(a_add, abuf ^ 1u, ja0 + (m >> 1));
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            if (m == (c8 == 0 ? 5 : 1) && c8 < BPW) {
                                __builtin_amdgcn_sched_barrier(0);
                                dma_b(sl3, tap3, buf3, c8);
                                __builtin_amdgcn_sched_barrier(0);
                            }
#pragma unroll
                            for (int r = 0; r < R; r++) {
                                if (m == 1 + (r * (M - 5)) / R) {         // reads spread over MFMAs 1 .. M-5: >= 4 MFMAs cover the LDS latency
                                    __builtin_amdgcn_sched_barrier(0);
                                    if (r < MB) a_nxt[r] = *a_addr(abuf_n, r, tap_n, c8_n);
                                    else b_nxt[r - MB] = *b_addr(bbuf_n, c8_n, r - MB);
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            }
                        }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mb = 0; mb < MB; mb++) a_cur[mb] = a_nxt[mb];
#pragma unroll
                for (int nb = 0; nb < NB; nb++) b_cur[nb] = b_nxt[nb];
            }
            // stage boundary.  The youngest transfer needed next is this wave's last piece of B(t+2),
            // issued in stage t-1 after that stage's A pieces; the only younger transfers are the
            // KA[t] + BPW issued in stage t (vmcnt retires in order; the epilogue stores of the
            // previous tile, if any, are older than stage t's transfers and are retired too).
            {
                // After an interior-tile epilogue exactly NST stores sit in the queue between B(t+2) and
                // stage t's transfers: counting them in lets the stores drain under the next two stages
                // instead of stalling the first one (n >= 63 = queue depth: nothing to wait for).
                const int n_dma = KA[tap] + BPW;
                if (EPI && tap == 0 && epi_stores) wait_vmcnt_n(n_dma + NST);
                else wait_vmcnt_n(n_dma);
                if (tap == 0) epi_stores = false;
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            gs++;
        }
        abuf ^= 1u;

        if (last_slice) {
            // ---- epilogue: bias + LeakyReLU, NHWC stores (C/D: column = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)).
            //      leaky(v) = max(v, 0.1f*v).  Interior tiles (all but the last row / column of tiles)
            //      take the unpredicated path. ----
            const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
            const int oy0 = tile_y * ROWS, ox0 = tile_x * 32;
            const bool interior = (oy0 + ROWS <= d.out_h) && (ox0 + 32 <= d.out_w);
            float *obase = d.out + (long long)(oy0 + wm * MB) * d.out_rs + (long long)(ox0 + 4 * kk) * COUT + nb0 * 32 + li;
            if (interior) {
#pragma unroll
                for (int mb = 0; mb < MB; mb++)
#pragma unroll
                    for (int nb = 0; nb < NB; nb++)
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const float v = acc[mb][nb][r] + bv[nb];
                            obase[(long long)mb * d.out_rs + ((r & 3) + 8 * (r >> 2)) * COUT + nb * 32] = __builtin_amdgcn_fmed3f(v, 0.1f * v, 3.402823466e+38f);
                            acc[mb][nb][r] = 0.0f;
                        }
                epi_stores = true;
            } else {
#pragma unroll
                for (int mb = 0; mb < MB; mb++) {
                    const int y = oy0 + wm * MB + mb;
#pragma unroll
                    for (int nb = 0; nb < NB; nb++)
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const int x = ox0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                            const float v = acc[mb][nb][r] + bv[nb];
                            if (y < d.out_h && x < d.out_w)
                                obase[(long long)mb * d.out_rs + ((r & 3) + 8 * (r >> 2)) * COUT + nb * 32] = __builtin_amdgcn_fmed3f(v, 0.1f * v, 3.402823466e+38f);
                            acc[mb][nb][r] = 0.0f;
                        }
                }
            }
            tile += per;
            if (tile >= chunk_end) break;
            sl = 0;
        } else {
            sl++;
        }
    }
    W2XC_WAIT_VMCNT(0);   // drain the speculative DMAs before the LDS is released
}

// ------------------------------------------------------------------------------------------------
// conv3x3_first: cin <= 3, planar input, cout = 32*NBT planes, NHWC output.
//   K = 9*CIN (k = c*9 + tap), padded to 2*S; lane (i, kk) feeds k = 2*s + kk of MFMA step s.
//   Packed weights: wpk[nb][s][lane] = W[32*nb + (lane&31)][c][tap] for k = 2*s + (lane>>5) < K, else 0.
//   Workgroup = 4 waves, tile = 8 rows x 32 pixels; wave w owns rows 2w, 2w+1.
// ------------------------------------------------------------------------------------------------
#ifndef FIRST_TPW
#define FIRST_TPW 4
#endif
template <int CIN, int NBT, bool PLANAR = false>
__global__ void __launch_bounds__(256) conv3x3_first(W2xcConvDesc d, int tiles_x, int ntiles)
{
    constexpr int ROWS = 8, MB = 2, HW = 34, HH = ROWS + 2;
    constexpr int K = 9 * CIN, S = (K + 1) / 2;
    constexpr int COUT = 32 * NBT;
    constexpr int TPS = 36;   // floats per pixel in the store-transpose tile: 32 planes + 4 pad (144-byte stride: conflict-free 16-byte writes)
    constexpr int PATCH = CIN * HH * HW, PL = (PATCH + 255) / 256;   // the tile's haloed source pixels; loads per thread
    __shared__ float lds2[2][PATCH];   // the tile's patch, double-buffered: the next tile's is written while this one's stores are still in flight
    __shared__ __attribute__((aligned(16))) float lbias[COUT];
    constexpr int PLS = 68;   // planar out: floats per plane in the store-transpose tile, MB rows x 32 pixels + 4 pad (272-byte stride: conflict-free 16-byte writes)
    __shared__ __attribute__((aligned(16))) float tps[PLANAR ? 4 * 32 * PLS : 4 * MB * 32 * TPS];   // per wave: its MB rows x 32 pixels x 32 planes

    // a workgroup walks FIRST_TPW consecutive tiles (a write-bound kernel of 32 791 four-wave workgroups was bound by their turnover)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kk = lane >> 5, i = lane & 31;
    const int tile_base = xcd_remap(blockIdx.x, (ntiles + FIRST_TPW - 1) / FIRST_TPW) * FIRST_TPW;
    // The layer's weights live in REGISTERS for the workgroup's lifetime (NBT x S <= 56 values per lane), the biases in LDS.  Round 6: loaded inside the
    // plane-block loop, the compiler gave all S of them ONE register -- fourteen L2 round trips per plane block, each behind s_waitcnt vmcnt(0) (which also
    // waited for the block's stores), and the patch fill was a loop of one load + vmcnt(0) per pass: 3 -> 128 on 2048 x 2048 ran 0.93 ms = 2.3 TB/s of
    // writes where the same store stream alone reaches 5.5 (tools/ubench/planar_store.hip -- the stores' shape was never the limit).
    float bw[NBT][S];
#pragma unroll
    for (int nb = 0; nb < NBT; nb++)
#pragma unroll
        for (int s = 0; s < S; s++) bw[nb][s] = d.wpk[(nb * S + s) * 64 + lane];
    for (int idx = threadIdx.x; idx < COUT; idx += 256) lbias[idx] = d.bias[idx];

    // the patch of one tile: PL loads per thread, all in flight at once (copyMakeBorder REPLICATE and INTER_NEAREST 2x folded into the addresses)
    auto patch_load = [&](int tile, float (&v)[PL]) {
        const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
        const int oy0 = tile_y * ROWS, ox0 = tile_x * 32;
#pragma unroll
        for (int t = 0; t < PL; t++) {
            const int idx = min(threadIdx.x + 256 * t, PATCH - 1);
            const int c = idx / (HH * HW), p = idx - c * (HH * HW);
            const int py = p / HW, px = p - py * HW;
            const int gy = clampi(oy0 + py + d.off_y, 0, d.in_h - 1) >> d.in_shift;
            const int gx = clampi(ox0 + px + d.off_x, 0, d.in_w - 1) >> d.in_shift;
            v[t] = d.in[(long long)c * d.in_cs + (long long)gy * d.in_rs + (long long)gx * d.in_ps];
        }
    };
    float pv[PL];
    if (tile_base < ntiles) patch_load(tile_base, pv);
    // (register VALUES from here on: an empty asm statement "uses" every weight, so the waits for their loads stand here and not -- conservatively,
    //  in every pass -- inside the tile loop)
#pragma unroll
    for (int nb = 0; nb < NBT; nb++)
#pragma unroll
        for (int s = 0; s < S; s++) asm volatile("" : "+v"(bw[nb][s]));
    auto patch_to_lds = [&](int buf) {
#pragma unroll
        for (int t = 0; t < PL; t++)
            if (threadIdx.x + 256 * t < PATCH) lds2[buf][threadIdx.x + 256 * t] = pv[t];
    };
    if (tile_base < ntiles) patch_to_lds(0);

  for (int it = 0; it < FIRST_TPW; it++) {
    const int tile = tile_base + it;
    if (tile >= ntiles) break;                       // (workgroup-uniform)
    const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
    const int oy0 = tile_y * ROWS, ox0 = tile_x * 32;
    // ONE barrier per tile: this tile's patch (written behind the previous tile's first plane block, below) is complete, and every wave has read
    // the patch of the tile before that, whose buffer the next write reuses
    __syncthreads();
    const float *lds = lds2[it & 1];
    const bool have_next = it + 1 < FIRST_TPW && tile + 1 < ntiles;

    float a[MB][S];
#pragma unroll
    for (int s = 0; s < S; s++) {
        const int k0 = 2 * s, k1 = 2 * s + 1;
        const int off0 = (k0 / 9) * (HH * HW) + ((k0 % 9) / 3) * HW + (k0 % 9) % 3;
        const int off1 = k1 < K ? (k1 / 9) * (HH * HW) + ((k1 % 9) / 3) * HW + (k1 % 9) % 3 : 0;
        const int off = kk ? off1 : off0;
#pragma unroll
        for (int mb = 0; mb < MB; mb++) a[mb][s] = lds[(wave * MB + mb) * HW + i + off];
    }
    // The next tile's patch is fetched under plane block 0's MFMAs and goes to LDS IN FRONT of that block's stores: vmcnt counts in order, so a wait
    // for these loads also waits for every store issued before them -- here the previous tile's, a whole tile old; at the tile's end it would be
    // this tile's own 32 stores, just issued (the compiler cannot count stores behind the edge tests and waits for all of them: measured, the
    // workgroup then runs store-acknowledge to store-acknowledge).
    if (have_next) patch_load(tile + 1, pv);

    if constexpr (PLANAR) {
        // planar out (the layout conv3x3_wino4 reads).  Pixels = MFMA A, weights = B: the accumulator tile is [pixel][plane], a lane owns plane 32 nb + i and per
        // register quad q the 4 CONSECUTIVE PIXELS 8q + 4kk .. +3 of a row.  The wave's MB rows x 32 pixels x 32 planes change owner through LDS (own region,
        // no workgroup barrier) and leave as 16-byte stores, 8 lanes = the 128-byte line of one (plane, row): 8 store instructions per plane block where the
        // dword form of rounds 3-5 (a half-wave = one line) issued 32 -- at ~28 cycles of the CU's address path per wave instruction those were the kernel's
        // time once the weight loads were out of the way (0.52 ms; stores alone in that shape: tools/ubench/planar_store.hip).
        float *tw = tps + wave * (32 * PLS);
#pragma unroll
        for (int nb = 0; nb < NBT; nb++) {
            const float bv = lbias[nb * 32 + i];
            f32x16 acc[MB];
#pragma unroll
            for (int mb = 0; mb < MB; mb++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[mb][r] = bv;
#pragma unroll
            for (int s = 0; s < S; s++)
#pragma unroll
                for (int mb = 0; mb < MB; mb++)
                    acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb][s], bw[nb][s], acc[mb], 0, 0, 0);
            if (nb == 0 && have_next) patch_to_lds((it + 1) & 1);
#pragma unroll
            for (int mb = 0; mb < MB; mb++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = leaky(acc[mb][4 * q + e]);
                    *reinterpret_cast<f32x4 *>(tw + i * PLS + mb * 32 + 8 * q + 4 * kk) = v;
                }
            // (16-byte stores want rows and planes on 16-byte boundaries and room for a row's last quad: the engine's workspaces have both -- rows of
            //  roundup32(w) floats -- and the launcher refuses anything else; the quad's columns beyond out_w hold finite values nobody reads)
            // lane = (pixel quad j, row mb, plane pl0 of a group of four); instruction n of a plane block covers planes 32 nb + 4 n + pl0
            const int j = lane & 7, mb = (lane >> 3) % MB, pl0 = (lane >> 3) / MB;
            static_assert(MB == 2, "lane map of the planar stores");
            const float *tr = tw + pl0 * PLS + mb * 32 + 4 * j;
            const int y = oy0 + wave * MB + mb, x = ox0 + 4 * j;
            float *ob = d.out + ((long long)pl0 * d.out_cs + (long long)y * d.out_rs + x);   // + a wave-uniform plane offset per instruction
#pragma unroll
            for (int n = 0; n < 8; n++) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(tr + n * 4 * PLS);
                if (y < d.out_h && x < d.out_w) *reinterpret_cast<f32x4 *>(ob + (long long)(nb * 32 + n * 4) * d.out_cs) = v;
            }
        }
        continue;   // (next tile)
    }

    // NHWC out.  Operands swapped (weights = MFMA A, pixels = B): the accumulator tile is [plane][pixel], a lane owns pixel ox0 + i and
    // per register quad q the 4 consecutive planes 32*nb + 8q + 4kk .. +3 -> one 16-byte store per quad instead of 4 scattered dwords;
    // the accumulators start at the bias.
#pragma unroll
    for (int nb = 0; nb < NBT; nb++) {
        f32x4 bq[4];
#pragma unroll
        for (int q = 0; q < 4; q++) bq[q] = *reinterpret_cast<const f32x4 *>(lbias + nb * 32 + 8 * q + 4 * kk);
        f32x16 acc[MB];
#pragma unroll
        for (int mb = 0; mb < MB; mb++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mb][r] = bq[r >> 2][r & 3];
#pragma unroll
        for (int s = 0; s < S; s++)
#pragma unroll
            for (int mb = 0; mb < MB; mb++)
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[nb][s], a[mb][s], acc[mb], 0, 0, 0);
        if (nb == 0 && have_next) patch_to_lds((it + 1) & 1);
        // Stores: a lane holds 4 x 4 consecutive planes of ONE pixel, so direct stores write 32-byte pieces of 32 different cache lines per
        // instruction -- 3.4-3.6 TB/s where a pure write stream reaches 6.9 (tools/ubench/hbm_streams.py).  The wave's MB x 32 pixels x 32
        // planes go through LDS instead (own region, no workgroup barrier) and leave as whole lines: 8 consecutive lanes = one pixel's
        // 128 bytes, 64 lanes = 8 pixels (1 KiB contiguous when COUT = 32).
        float *tw = tps + wave * (MB * 32 * TPS);
#pragma unroll
        for (int mb = 0; mb < MB; mb++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = leaky(acc[mb][4 * q + e]);
                *reinterpret_cast<f32x4 *>(tw + (mb * 32 + i) * TPS + 8 * q + 4 * kk) = v;
            }
#pragma unroll
        for (int n = 0; n < MB * 4; n++) {
            const int c = n * 64 + lane;             // 16-byte chunk c of the wave's MB x 32 x 8 chunks
            const int p = c >> 3, ch = c & 7;         // pixel p = mb * 32 + x, chunk ch of its 32 planes
            const int mb = p >> 5, px = p & 31;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(tw + p * TPS + 4 * ch);
            const int y = oy0 + wave * MB + mb, x = ox0 + px;
            if (y < d.out_h && x < d.out_w)
                *reinterpret_cast<f32x4 *>(d.out + (long long)y * d.out_rs + (long long)x * COUT + nb * 32 + 4 * ch) = v;
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// conv3x3_last: cout <= 3, NHWC input, planar output.  "Taps as N".
//   For each haloed pixel q of the tile: G[q][n] = sum_c in[q][c] * W[o][c][tap], n = tap*COUT + o,
//   N = 9*COUT padded to 16*NB16, on v_mfma_f32_16x16x4_f32 (M = 16 pixels, K = 4).  A comes
//   straight from global memory: lane (i = lane&15, kk = lane>>4) loads channels
//   16*s4 + 4*kk .. +3 of pixel i as one dwordx4; MFMA j of group s4 contracts channels
//   {16*s4 + 4*kk + j}.  Packed weights: wpk[s4][j][nb][lane] = W[o][16*s4 + 4*(lane>>4) + j][tap]
//   for n = 16*nb + (lane&15) = tap*COUT + o < 9*COUT, else 0.
//   Then out[p][o] = leaky(bias[o] + sum_tap G[p + tap][tap*COUT + o]) via LDS.
//   Workgroup = 4 waves, tile = ROWS x 32 output pixels.
// ------------------------------------------------------------------------------------------------
#ifndef LAST_TPW
#define LAST_TPW 4
#endif
template <int CIN, int COUT>
__global__ void __launch_bounds__(256) conv3x3_last(W2xcConvDesc d, int tiles_x, int ntiles)
{
    constexpr int ROWS = 8, HW = 34, HH = ROWS + 2, NPIX = HH * HW;
    constexpr int NBLK = (NPIX + 15) / 16;
    constexpr int N = 9 * COUT, NB16 = (N + 15) / 16;
    constexpr int GS = N | 1;                  // odd LDS row stride
    constexpr int S4 = CIN / 16;
    __shared__ float G[NBLK * 16 * GS];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kk = lane >> 4, i = lane & 15;

    // the layer's weights once per workgroup, which walks LAST_TPW consecutive tiles (round 6: one tile per workgroup re-read the 16 KiB image per tile,
    // and a wave's pixel blocks ran load -> wait -> 64 MFMAs one after the other: 3.9 TB/s of reads on 128 -> 3)
    float b[S4][4][NB16];
#pragma unroll
    for (int s4 = 0; s4 < S4; s4++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int nb = 0; nb < NB16; nb++) b[s4][j][nb] = d.wpk[((s4 * 4 + j) * NB16 + nb) * 64 + lane];

    // the 16 channels-of-four of pixel block `blk` of tile `tile` (clamped: the haloed tile's pixels outside the plane repeat the edge)
    auto blk_load = [&](int tile, int blk, f32x4 (&v)[S4]) {
        const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
        int q = blk * 16 + i;
        q = q < NPIX ? q : NPIX - 1;
        const int py = q / HW, px = q - py * HW;
        const int gy = clampi(tile_y * ROWS + py + d.off_y, 0, d.in_h - 1);
        const int gx = clampi(tile_x * 32 + px + d.off_x, 0, d.in_w - 1);
        const f32x4 *src = reinterpret_cast<const f32x4 *>(d.in + (long long)gy * d.in_rs + (long long)gx * CIN) + kk;
#pragma unroll
        for (int s4 = 0; s4 < S4; s4++) v[s4] = src[s4 * 4];
    };

    float bo[COUT];
#pragma unroll
    for (int o = 0; o < COUT; o++) bo[o] = d.bias[o];
    const int tile_base = xcd_remap(blockIdx.x, (ntiles + LAST_TPW - 1) / LAST_TPW) * LAST_TPW;
    f32x4 av[S4];
    if (tile_base < ntiles) blk_load(tile_base, wave, av);
    // (register VALUES from here on: left pending, the weights' waits sit INSIDE the loop -- s_waitcnt vmcnt(7) behind the eight loads of the next
    //  block, i.e. a wait for the first of those -- in every pass)
#pragma unroll
    for (int s4 = 0; s4 < S4; s4++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int nb = 0; nb < NB16; nb++) asm volatile("" : "+v"(b[s4][j][nb]));
    for (int it = 0; it < LAST_TPW; it++) {
        const int tile = tile_base + it;
        if (tile >= ntiles) break;                       // (workgroup-uniform)
        const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
        const int oy0 = tile_y * ROWS, ox0 = tile_x * 32;
        for (int blk = wave; blk < NBLK; blk += 4) {
            // the next block's loads (this tile's, or the first of the next tile) fly under this block's MFMAs
            f32x4 nv[S4];
            const bool more = blk + 4 < NBLK;
            const bool next_tile = !more && it + 1 < LAST_TPW && tile + 1 < ntiles;
            if (more) blk_load(tile, blk + 4, nv);
            else if (next_tile) blk_load(tile + 1, wave, nv);
            f32x4 acc[NB16];
#pragma unroll
            for (int nb = 0; nb < NB16; nb++) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < S4; s4++)
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int nb = 0; nb < NB16; nb++)
                        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4][j], b[s4][j][nb], acc[nb], 0, 0, 0);
            // C/D map of 16x16 MFMA: column = lane&15 (n), row = 4*(lane>>4) + r (pixel in block)
#pragma unroll
            for (int nb = 0; nb < NB16; nb++) {
                const int n = nb * 16 + i;
                if (n < N) {
#pragma unroll
                    for (int r = 0; r < 4; r++) G[(blk * 16 + kk * 4 + r) * GS + n] = acc[nb][r];
                }
            }
            if (more || next_tile) {
#pragma unroll
                for (int s4 = 0; s4 < S4; s4++) av[s4] = nv[s4];
            }
        }
        __syncthreads();

        for (int p = threadIdx.x; p < ROWS * 32; p += 256) {
            const int py = p >> 5, px = p & 31;
            const int y = oy0 + py, x = ox0 + px;
            if (y >= d.out_h || x >= d.out_w) continue;
#pragma unroll
            for (int o = 0; o < COUT; o++) {
                float v = 0.0f;
#pragma unroll
                for (int tap = 0; tap < 9; tap++)
                    v += G[((py + tap / 3) * HW + px + tap % 3) * GS + tap * COUT + o];
                d.out[(long long)o * d.out_cs + (long long)y * d.out_rs + (long long)x * d.out_ps] = leaky(v + bo[o]);
            }
        }
        __syncthreads();                                 // (G is rewritten by the next tile)
    }
}

// replicate-padded planar copy: dst plane c (ph x pw pixels, row stride d_rs, plane stride d_cs) = src(clamp(y - pad), clamp(x - pad)) -- the
// BORDER_REPLICATE of Model::filter (modelHandler.cpp:141-142) made explicit for the kernels that run a valid conv on aligned planar tiles
__global__ void __launch_bounds__(256) pad_planar_kernel(const float *src, long long s_rs, long long s_ps, long long s_cs, float *dst, long long d_rs,
                                                         long long d_cs, int h, int w, int c, int pad)
{
    const int pw = w + 2 * pad, ph = h + 2 * pad;
    const long long total = (long long)c * ph * pw;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int x = (int)(idx % pw);
        const long long r = idx / pw;
        const int y = (int)(r % ph), ch = (int)(r / ph);
        const int sy = clampi(y - pad, 0, h - 1), sx = clampi(x - pad, 0, w - 1);
        dst[(long long)ch * d_cs + (long long)y * d_rs + x] = src[(long long)ch * s_cs + (long long)sy * s_rs + (long long)sx * s_ps];
    }
}

// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) repack_kernel(const float *src, long long s_rs, long long s_ps, long long s_cs,
                                                     float *dst, long long d_rs, long long d_ps, long long d_cs,
                                                     int h, int w, int c)
{
    const long long total = (long long)h * w * c;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int ch = (int)(idx % c);
        const long long p = idx / c;
        const int x = (int)(p % w), y = (int)(p / w);
        dst[y * d_rs + x * d_ps + ch * d_cs] = src[y * s_rs + x * s_ps + ch * s_cs];
    }
}

// ================================================================================================
// host side: kernel selection, weight packing, launch
// ================================================================================================
static bool is_mid(int c) { return c == 32 || c == 64 || c == 128; }

W2xcKernelKind w2xc_pick_kernel(int cin, int cout)
{
    if (is_mid(cin) && is_mid(cout)) return W2XC_K_MFMA;
    if ((cin == 1 || cin == 3) && is_mid(cout)) return W2XC_K_FIRST;
    if (is_mid(cin) && (cout == 1 || cout == 3)) return W2XC_K_LAST;
    return W2XC_K_DIRECT;
}

const char *w2xc_kernel_name(W2xcKernelKind kind, int cin, int cout)
{
    (void)cin; (void)cout;
    switch (kind) {
    case W2XC_K_MFMA: return "conv3x3_mfma";
    case W2XC_K_FIRST: return "conv3x3_first";
    case W2XC_K_LAST: return "conv3x3_last";
    case W2XC_K_MID_SPLIT: return "conv3x3_split";
    case W2XC_K_FIRST_SPLIT: return "conv3x3_first_split";
    case W2XC_K_LAST_GATHER: return "conv3x3_last_gather";
    case W2XC_K_FIRST2_SPLIT: return "conv3x3_first2_split";
    case W2XC_K_FUSED_AWAY: return "(in_next_layer)";
    case W2XC_K_FIRST2_WINO4: return "conv3x3_first2_wino4";
    default: return "conv3x3_direct";
    }
}

static int direct_cout_pad(int cout) { return (cout + DIRECT_CG - 1) / DIRECT_CG * DIRECT_CG; }

size_t w2xc_packed_weight_floats(W2xcKernelKind kind, int cin, int cout)
{
    switch (kind) {
    case W2XC_K_MFMA: return (size_t)9 * cin * cout;
    case W2XC_K_FIRST: case W2XC_K_FIRST_SPLIT: return (size_t)(cout / 32) * ((9 * cin + 1) / 2) * 64;
    case W2XC_K_LAST: return (size_t)(cin / 16) * 4 * ((9 * cout + 15) / 16) * 64;
    default: return (size_t)cin * 9 * direct_cout_pad(cout);
    }
}

void w2xc_pack_weights(W2xcKernelKind kind, int cin, int cout, const float *w, float *dst)
{
    auto W = [&](int o, int i, int tap) { return w[((size_t)o * cin + i) * 9 + tap]; };
    memset(dst, 0, w2xc_packed_weight_floats(kind, cin, cout) * sizeof(float));
    if (kind == W2XC_K_FIRST_SPLIT) kind = W2XC_K_FIRST;
    if (kind == W2XC_K_MFMA) {
        const int nbt = cout / 32, c8n = cin / 8;
        for (int tap = 0; tap < 9; tap++)
            for (int c8 = 0; c8 < c8n; c8++)
                for (int nb = 0; nb < nbt; nb++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int j = 0; j < 4; j++) {
                            const int kk = lane >> 5, n = lane & 31;
                            dst[((((size_t)tap * c8n + c8) * nbt + nb) * 64 + lane) * 4 + j] =
                                W(nb * 32 + n, c8 * 8 + kk * 4 + j, tap);
                        }
    } else if (kind == W2XC_K_FIRST) {
        const int nbt = cout / 32, K = 9 * cin, S = (K + 1) / 2;
        for (int nb = 0; nb < nbt; nb++)
            for (int s = 0; s < S; s++)
                for (int lane = 0; lane < 64; lane++) {
                    const int k = 2 * s + (lane >> 5);
                    dst[((size_t)nb * S + s) * 64 + lane] = k < K ? W(nb * 32 + (lane & 31), k / 9, k % 9) : 0.0f;
                }
    } else if (kind == W2XC_K_LAST) {
        const int N = 9 * cout, nb16 = (N + 15) / 16, s4n = cin / 16;
        for (int s4 = 0; s4 < s4n; s4++)
            for (int j = 0; j < 4; j++)
                for (int nb = 0; nb < nb16; nb++)
                    for (int lane = 0; lane < 64; lane++) {
                        const int n = nb * 16 + (lane & 15), c = 16 * s4 + 4 * (lane >> 4) + j;
                        dst[(((size_t)s4 * 4 + j) * nb16 + nb) * 64 + lane] = n < N ? W(n % cout, c, n / cout) : 0.0f;
                    }
    } else {
        const int cp = direct_cout_pad(cout);
        for (int i = 0; i < cin; i++)
            for (int tap = 0; tap < 9; tap++)
                for (int o = 0; o < cout; o++) dst[((size_t)i * 9 + tap) * cp + o] = W(o, i, tap);
    }
}

template <int CIN, int COUT, int MB, int NB, int WM, int WN>
static hipError_t launch_mfma2(const W2xcConvDesc &d, hipStream_t stream)
{
    const int tiles_x = (d.out_w + 31) / 32, tiles_y = (d.out_h + 7) / 8;
    const int ntiles = tiles_x * tiles_y;
    constexpr int NW = WM * WN;
    const size_t lds_bytes = 2 * (size_t)(NW * ((43 + NW) / NW) * 1024) + 4 * (size_t)(4 * (COUT / 32) * 1024);
    auto kern = conv3x3_mfma2<CIN, COUT, MB, NB, WM, WN>;
    // > 64 KiB of dynamic LDS needs the opt-in attribute, and function attributes are per DEVICE
    // (the in-process multi-GPU path launches this kernel on several devices from several threads)
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 64 || !((attr_done.load() >> dev) & 1ull)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        if (dev < 64) attr_done.fetch_or(1ull << dev);
    }
    int grid = 256;   // one persistent workgroup per CU; a multiple of 8 (one share per XCD)
    if (grid > ((ntiles + 7) & ~7)) grid = (ntiles + 7) & ~7;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds_bytes, stream, d, tiles_x, ntiles);
    return hipGetLastError();
}

template <typename KernelT>
static hipError_t launch_tiled8(KernelT kernel, const W2xcConvDesc &d, hipStream_t stream)
{
    const int tiles_x = (d.out_w + 31) / 32, tiles_y = (d.out_h + 7) / 8;
    const int ntiles = tiles_x * tiles_y;
    hipLaunchKernelGGL(kernel, dim3(ntiles), dim3(256), 0, stream, d, tiles_x, ntiles);
    return hipGetLastError();
}

template <typename KernelT>
static hipError_t launch_last(KernelT kernel, const W2xcConvDesc &d, hipStream_t stream)
{
    const int tiles_x = (d.out_w + 31) / 32, tiles_y = (d.out_h + 7) / 8;
    const int ntiles = tiles_x * tiles_y;
    hipLaunchKernelGGL(kernel, dim3((ntiles + LAST_TPW - 1) / LAST_TPW), dim3(256), 0, stream, d, tiles_x, ntiles);
    return hipGetLastError();
}

template <typename KernelT>
static hipError_t launch_first(KernelT kernel, const W2xcConvDesc &d, hipStream_t stream)
{
    const int tiles_x = (d.out_w + 31) / 32, tiles_y = (d.out_h + 7) / 8;
    const int ntiles = tiles_x * tiles_y;
    hipLaunchKernelGGL(kernel, dim3((ntiles + FIRST_TPW - 1) / FIRST_TPW), dim3(256), 0, stream, d, tiles_x, ntiles);
    return hipGetLastError();
}

hipError_t w2xc_launch_conv(W2xcKernelKind kind, const W2xcConvDesc &d, hipStream_t stream)
{
    if (d.out_w <= 0 || d.out_h <= 0) return hipSuccess;
    if (d.in_shift != 0 && kind != W2XC_K_FIRST && kind != W2XC_K_DIRECT) return hipErrorInvalidValue;
    if (kind == W2XC_K_MFMA) {
        if (d.in_ps != d.cin || d.in_cs != 1 || d.out_ps != d.cout || d.out_cs != 1) return hipErrorInvalidValue;
        const int key = d.cin * 1000 + d.cout;
        // Measured inside the 7-layer model (round 2, same box, same run): 8 waves (two per SIMD: the partner's MFMAs cover
        // this wave's non-MFMA issue slots) win wherever the output has >= 64 planes (32->64 -1.8 %, 64->64 -2.5 %,
        // 64->128 -2.9 %, 128->128 -2.9 %); 32->32 has one plane block, so its 8 waves would split rows only.
        const bool w8 = d.cout >= 64;
        switch (key) {
        //                                  CIN  COUT  MB NB WM WN
        case 32032:  return launch_mfma2<32, 32, 2, 1, 4, 1>(d, stream);
        case 32064:  return w8 ? launch_mfma2<32, 64, 2, 1, 4, 2>(d, stream) : launch_mfma2<32, 64, 2, 2, 4, 1>(d, stream);
        case 32128:  return w8 ? launch_mfma2<32, 128, 2, 2, 4, 2>(d, stream) : launch_mfma2<32, 128, 4, 2, 2, 2>(d, stream);
        case 64032:  return launch_mfma2<64, 32, 2, 1, 4, 1>(d, stream);
        case 64064:  return w8 ? launch_mfma2<64, 64, 2, 1, 4, 2>(d, stream) : launch_mfma2<64, 64, 2, 2, 4, 1>(d, stream);
        case 64128:  return w8 ? launch_mfma2<64, 128, 2, 2, 4, 2>(d, stream) : launch_mfma2<64, 128, 4, 2, 2, 2>(d, stream);
        case 128032: return launch_mfma2<128, 32, 2, 1, 4, 1>(d, stream);
        case 128064: return w8 ? launch_mfma2<128, 64, 2, 1, 4, 2>(d, stream) : launch_mfma2<128, 64, 2, 2, 4, 1>(d, stream);
        case 128128: return w8 ? launch_mfma2<128, 128, 2, 2, 4, 2>(d, stream) : launch_mfma2<128, 128, 4, 2, 2, 2>(d, stream);
        default: return hipErrorInvalidValue;
        }
    }
    if (kind == W2XC_K_FIRST && d.out_ps == 1 && d.cout > 1) {   // planar out: whole pixel quads (the engine's planar workspaces; Model::filter goes through NHWC)
        if (((d.out_rs | d.out_cs) & 3) != 0 || d.out_rs < ((d.out_w + 3) & ~3) || (((size_t)d.out) & 15) != 0) return hipErrorInvalidValue;
        switch (d.cin * 1000 + d.cout) {
        case 1032: return launch_first(conv3x3_first<1, 1, true>, d, stream);
        case 1064: return launch_first(conv3x3_first<1, 2, true>, d, stream);
        case 1128: return launch_first(conv3x3_first<1, 4, true>, d, stream);
        case 3032: return launch_first(conv3x3_first<3, 1, true>, d, stream);
        case 3064: return launch_first(conv3x3_first<3, 2, true>, d, stream);
        case 3128: return launch_first(conv3x3_first<3, 4, true>, d, stream);
        default: return hipErrorInvalidValue;
        }
    }
    if (kind == W2XC_K_FIRST) {
        if (d.out_ps != d.cout || d.out_cs != 1) return hipErrorInvalidValue;
        const int key = d.cin * 1000 + d.cout;
        switch (key) {
        case 1032: return launch_first(conv3x3_first<1, 1>, d, stream);
        case 1064: return launch_first(conv3x3_first<1, 2>, d, stream);
        case 1128: return launch_first(conv3x3_first<1, 4>, d, stream);
        case 3032: return launch_first(conv3x3_first<3, 1>, d, stream);
        case 3064: return launch_first(conv3x3_first<3, 2>, d, stream);
        case 3128: return launch_first(conv3x3_first<3, 4>, d, stream);
        default: return hipErrorInvalidValue;
        }
    }
    if (kind == W2XC_K_LAST) {
        if (d.in_ps != d.cin || d.in_cs != 1) return hipErrorInvalidValue;
        const int key = d.cin * 1000 + d.cout;
        switch (key) {
        case 32001:  return launch_last(conv3x3_last<32, 1>, d, stream);
        case 64001:  return launch_last(conv3x3_last<64, 1>, d, stream);
        case 128001: return launch_last(conv3x3_last<128, 1>, d, stream);
        case 32003:  return launch_last(conv3x3_last<32, 3>, d, stream);
        case 64003:  return launch_last(conv3x3_last<64, 3>, d, stream);
        case 128003: return launch_last(conv3x3_last<128, 3>, d, stream);
        default: return hipErrorInvalidValue;
        }
    }
    const int cp = direct_cout_pad(d.cout);
    dim3 grid((d.out_w + 63) / 64, (d.out_h + 3) / 4, cp / DIRECT_CG);
    hipLaunchKernelGGL(conv3x3_direct, grid, dim3(64, 4), 0, stream, d, cp);
    return hipGetLastError();
}

hipError_t w2xc_launch_pad_planar(const float *src, long long s_rs, long long s_ps, long long s_cs, float *dst, long long d_rs, long long d_cs,
                                  int h, int w, int c, int pad, hipStream_t stream)
{
    const long long total = (long long)c * (h + 2 * pad) * (w + 2 * pad);
    if (total <= 0) return hipSuccess;
    long long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(pad_planar_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, s_rs, s_ps, s_cs, dst, d_rs, d_cs, h, w, c, pad);
    return hipGetLastError();
}

hipError_t w2xc_launch_repack(const float *src, long long s_rs, long long s_ps, long long s_cs, float *dst,
                              long long d_rs, long long d_ps, long long d_cs, int h, int w, int c, hipStream_t stream)
{
    const long long total = (long long)h * w * c;
    if (total <= 0) return hipSuccess;
    long long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(repack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, s_rs, s_ps, s_cs, dst, d_rs,
                       d_ps, d_cs, h, w, c);
    return hipGetLastError();
}
