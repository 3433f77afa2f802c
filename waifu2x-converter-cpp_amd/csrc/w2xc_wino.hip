// w2xc_wino.hip -- conv3x3_wino: the 3x3 x Cin x Cout contraction of Model::filterWorker
// (/root/reference/src/modelHandler.cpp:117-159) as Winograd F(2x2, 3x3) on the fp32 MFMA of gfx950.
//
// CDNA4's fp32 MFMA runs at the fp32 VECTOR rate (157 TFLOP/s): layers 2..6 of the 7-layer model are bound by it and
// conv3x3_mfma2 already holds 92 % of it.  What is left is doing fewer multiplies: for a 2x2 block of outputs
//
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A            g = 3x3 taps, d = 4x4 input patch, (.) = element-wise
//
// needs 16 multiplies per (output plane, input plane) instead of 36 -- 2.25x fewer MFMAs -- and everything else is
// additions in fp32 (all transform coefficients are 0, +-1, +-1/2: exact), so this is still fp32 arithmetic with a
// different summation order (measured against the CPU oracle in tests/: same 1e-4 gate as conv3x3_mfma2).
// The 16 "positions" xi of the transformed domain are 16 independent GEMMs  M_xi[o][t] = sum_c U_xi[o][c] V_xi[c][t]
// (t = 2x2 output block), each on v_mfma_f32_32x32x2_f32 with U = weights as the A operand and V as the B operand:
// lane (t, kk) TRANSFORMS ITS OWN PATCH in registers (32 additions per 16 MFMAs) -- V never exists in memory --
// and owns the 16 x 16 accumulators of its block column (256 AGPRs, all of them).
//
//   Workgroup  4 waves (one per SIMD, 512 registers), persistent, one per CU.  Work item = (16 rows x 32 pixels of output,
//              one block of 32 output planes); the Cout/32 items of a pixel tile are neighbours in the list, so they
//              run at the same time on CUs of one XCD and share the input tile in that XCD's L2.
//   Wave w     rows 4w .. 4w+3 of the tile = 2 x 16 blocks of 2x2 = the 32 columns of its MFMAs.
//   Stage      one 16-channel slice of the input: lane half kk works on channels 8kk .. 8kk+7 of the slice, two per
//              k-group (one ds_read_b64 per patch pixel), 8 steps of 16 MFMAs (one per xi) = 128 MFMAs = 8192 cycles.
//   LDS        A[2] x 40 KiB: the 18 x 34 pixel halo tile of a slice, 64 B per pixel, by LDS-DMA.  Pixels of a row are
//              stored even columns first, then odd ones, and the four 16-byte chunks of a pixel are XOR-swizzled with
//              bits 2..3 of the pixel index, so the 32 lanes that read patch element (r, c) of horizontally adjacent
//              blocks (2 pixels apart) spread over the banks.
//              B[2] x 32 KiB: U of one (plane block, slice) in fragment order [k-group][step][xi/4][lane][xi%4].
//              + 10 KiB: the per-lane source offsets of the 10 tile pieces a wave transfers per slice (registers are the scarce
//              resource: 256 accumulators + ~200 VGPRs), + the bias vector.
//   Epilogue   output transform (24 additions per plane), bias, LeakyReLU, 16-byte NHWC stores.
//   Banding    the 2x2 blocks sit on EVEN rows of the layer's whole output (W2xcConvDesc::wino_py): results do not depend on the band origin.
// Measured (round 2, 2160x3840, profiles/): 128->128 10.4 ms vs 16.9 ms for conv3x3_mfma2 -- 235 TFLOP/s of algorithmic FLOPs, 1.5x the
// MFMA roofline of a direct convolution, 2/3 of the MFMA peak on the multiplies it really issues; what the rest is, DESIGN.md 3.
#include "w2xc_kernels.h"
#include "w2xc_device.h"

#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <type_traits>

typedef float f32x2v __attribute__((ext_vector_type(2)));

template <int CIN, int COUT>
__global__ void __launch_bounds__(256, 1) conv3x3_wino(W2xcConvDesc d, int tiles_x, int nitems)
{
    constexpr int ROWS = 16, HW = 34, HH = ROWS + 2, NPIX = HH * HW;   // 612 halo pixels
    constexpr int NSL = CIN / 16, NOB = COUT / 32;
    constexpr int NW = 4;
    constexpr int A_SLOTS = NPIX * 4;                      // 16-byte slots of one slice
    constexpr int APW = (A_SLOTS + NW * 64 - 1) / (NW * 64);   // 10 pieces (1 KiB) per wave per slice
    constexpr unsigned A_BYTES = NW * APW * 1024;          // 40 KiB
    constexpr unsigned B_BYTES = 32 * 1024;                // U of one (plane block, slice)
    constexpr unsigned B_BASE = 2 * A_BYTES;
    static_assert(CIN % 16 == 0 && COUT % 32 == 0, "planes");
    // pixel tiles are walked in strips of 32 tiles, row by row inside a strip (the next round of an XCD is the tile
    // row below, whose halo rows are still in that XCD's L2)
    constexpr int STRIP = 32;
    const int tiles_y = nitems / (NOB * tiles_x);
    auto tile_coords = [&](int pt, int &ty_, int &tx_) {
        const int per_strip = STRIP * tiles_y;
        int sidx = pt / per_strip;
        const int nfull = tiles_x / STRIP;
        if (sidx > nfull) sidx = nfull;
        const int wid = sidx < nfull ? STRIP : tiles_x - nfull * STRIP;
        const int q = pt - sidx * per_strip;
        ty_ = q / wid;
        tx_ = sidx * STRIP + (q - ty_ * wid);
    };
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    const char *ldsb = reinterpret_cast<const char *>(lds);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, kk = lane >> 5;
    const int tyl = n >> 4, tx = n & 15;                   // this lane's 2x2 block inside the wave's 2 x 16 blocks

    // persistent schedule: XCD x (= blockIdx % 8) walks its own contiguous chunk of the item list
    const int xcd = blockIdx.x & 7, per = gridDim.x >> 3;
    const int cq = nitems >> 3, cr = nitems & 7;
    const int chunk_begin = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
    const int chunk_end = chunk_begin + cq + (xcd < cr ? 1 : 0);
    int item = chunk_begin + (blockIdx.x >> 3);
    if (item >= chunk_end) return;

    // ---- per-lane DMA source offsets of the halo tile (16-byte units) ----
    const f32x4 *in4 = reinterpret_cast<const f32x4 *>(d.in);
    unsigned goff[APW];
    auto slot_of = [&](int jj, int &row, int &col, int &q) {
        int s = (jj * NW + wave) * 64 + lane;
        s = s < A_SLOTS ? s : A_SLOTS - 1;                 // slots past the tile re-read its last one
        const int pp = s >> 2;                             // position of the pixel in the even-then-odd row order
        q = (s & 3) ^ ((pp >> 2) & 3);                     // chunk stored at this slot
        row = pp / HW;
        const int rem = pp - row * HW;
        col = rem < 17 ? 2 * rem : 2 * (rem - 17) + 1;
    };
    // tile-independent part of the offsets (interior tiles: + one base): per-lane constants, parked in LDS (10 KiB behind the U ring)
    // rather than in 10 registers -- the stage needs every VGPR, and the clamped path costs ~10k cycles per item when taken each time
    constexpr unsigned LOFS_BASE = B_BASE + 2 * B_BYTES;
    constexpr unsigned BIAS_BASE = LOFS_BASE + APW * 256 * 4;   // bias[COUT]: read in every item's epilogue (a global load there is an exposed round trip)
    for (int c = threadIdx.x; c < COUT; c += 256) lds[BIAS_BASE / 4 + c] = d.bias[c];   // (visible after the prologue barrier)
    {
        unsigned *lofs = reinterpret_cast<unsigned *>(const_cast<char *>(ldsb) + LOFS_BASE);
#pragma unroll
        for (int jj = 0; jj < APW; jj++) {
            int row, col, q;
            slot_of(jj, row, col, q);
            lofs[jj * 256 + threadIdx.x] = (unsigned)(((long long)row * d.in_rs + (long long)col * CIN) >> 2) + q;
        }
    }
    auto tile_offsets = [&](int it) {
        const int pt = it / NOB;
        int ty_, tx_;
        tile_coords(pt, ty_, tx_);
        const int y0 = ty_ * ROWS - d.wino_py + d.off_y, x0 = tx_ * 32 + d.off_x;
        if (y0 >= 0 && y0 + HH <= d.in_h && x0 >= 0 && x0 + HW <= d.in_w) {   // wave-uniform
            const unsigned base = (unsigned)(((long long)y0 * d.in_rs + (long long)x0 * CIN) >> 2);
            const unsigned *lofs = reinterpret_cast<const unsigned *>(ldsb + LOFS_BASE);
#pragma unroll
            for (int jj = 0; jj < APW; jj++) goff[jj] = lofs[jj * 256 + threadIdx.x] + base;
            return;
        }
#pragma unroll
        for (int jj = 0; jj < APW; jj++) {
            int row, col, q;
            slot_of(jj, row, col, q);
            const int gy = clampi(y0 + row, 0, d.in_h - 1);
            const int gx = clampi(x0 + col, 0, d.in_w - 1);
            goff[jj] = (unsigned)(((long long)gy * d.in_rs + (long long)gx * CIN) >> 2) + q;
        }
    };
    auto dma_a = [&](unsigned add, unsigned abuf, int jj) {
        lds_dma16(in4 + goff[jj] + add, lds0 + abuf * A_BYTES + (unsigned)(jj * NW + wave) * 1024u);
    };
    // U of (plane block ob, slice sl): 32 pieces of 1 KiB, 8 per wave
    const unsigned b_voff = (unsigned)lane * 16u;
    auto dma_b = [&](int ob, int sl_, unsigned buf, int jb) {
        const char *sbase = reinterpret_cast<const char *>(d.wpk) + ((size_t)(ob * NSL + sl_) * 32 + wave * 8 + (jb & 4)) * 1024;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(lds0 + B_BASE + buf * B_BYTES + (unsigned)(wave * 8 + (jb & 4)) * 1024u);
        switch (jb & 3) {
        case 0: lds_dma16_s<0>(sbase, b_voff, dst); break;
        case 1: lds_dma16_s<1024>(sbase, b_voff, dst); break;
        case 2: lds_dma16_s<2048>(sbase, b_voff, dst); break;
        default: lds_dma16_s<3072>(sbase, b_voff, dst); break;
        }
    };

    // ---- fragment addressing ----
    // patch element (r, c) of this lane's block: halo pixel (4w + 2tyl + r, 2tx + c), stored at position
    //   pp = row*34 + (col >> 1) + 17*(col & 1)   of the even-then-odd order; chunk q of it at byte pp*64 + ((q ^ ((pp>>2)&3)) << 4).
    // This lane half reads chunk q = 2kk + (G >> 1), 8-byte half G & 1:   a1[r][c] ^ ((G >> 1) << 4), + 8*(G & 1)
    unsigned a1[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int pp = (4 * wave + 2 * tyl + r) * HW + tx + (c >> 1) + 17 * (c & 1);
            a1[r][c] = (unsigned)(pp * 64) | ((unsigned)(((pp >> 2) & 3) ^ (2 * kk)) << 4);
        }

    // ---- prologue: A(slice 0) and U(slice 0) of the first item ----
    tile_offsets(item);
#pragma unroll
    for (int jj = 0; jj < APW; jj++) dma_a(0, 0, jj);
#pragma unroll
    for (int jb = 0; jb < 8; jb++) dma_b(item % NOB, 0, 0, jb);
    W2XC_WAIT_VMCNT(0);
    __builtin_amdgcn_s_barrier();

    unsigned abuf = 0, bbuf = 0;
    // (a trip count the compiler cannot see: with NSL = 2 it otherwise peels the one-iteration slice loop into the item loop and spills 31 registers)
    const int nsl_rt = NSL + (d.in_shift & 0x40000000);
    for (;;) {
      // The accumulators are DEFINED by the first stage of an item (its first 16 MFMAs take C = 0) and die in the epilogue: carried
      // across items they are 256 loop-carried registers whose phi copies the allocator routes through VGPRs and scratch.
      f32x16 acc[16];
      auto stage = [&](auto FIRST, int sl) {
        constexpr bool first = decltype(FIRST)::value;
        const bool last_slice = (sl == NSL - 1);
        const int item_n = item + per < chunk_end ? item + per : item;   // (the last item prefetches itself: harmless)
        unsigned a_add = (unsigned)(sl + 1) * 4;
        int ob_n = item % NOB, sl_n = sl + 1;
        if (last_slice) {
            tile_offsets(item_n);
            a_add = 0;
            ob_n = item_n % NOB;
            sl_n = 0;
        }

        // ---- one stage: 4 k-groups x 2 steps x 16 MFMAs ----
        f32x2v raw[16];
        f32x4 u_c[4], u_n[4];
        // V = B^T d B (B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]) for BOTH channels of a k-group at once, on the 2-vectors the
        // patch reads deliver (v_pk_add_f32: 32 per k-group): v2[kg & 1][xi][s] is the B operand of MFMA xi of step (kg, s).
        f32x2v v2[2][16];
        f32x2v tq2[4][4];
        auto load_raw = [&](int G) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int c = 0; c < 4; c++)
                    raw[r * 4 + c] = *reinterpret_cast<const f32x2v *>(ldsb + abuf * A_BYTES + (a1[r][c] ^ (unsigned)((G >> 1) << 4)) + 8 * (G & 1));
        };
        auto load_u = [&](f32x4 (&u)[4], int G, int s) {
#pragma unroll
            for (int q = 0; q < 4; q++)
                u[q] = *reinterpret_cast<const f32x4 *>(ldsb + B_BASE + bbuf * B_BYTES + (((G * 2 + s) * 4 + q) * 64 + lane) * 16);
        };
        auto transform_cols = [&](int k) {   // columns of d first (tq2), then rows
            const f32x2v d0 = raw[0 * 4 + k], d1 = raw[1 * 4 + k], d2 = raw[2 * 4 + k], d3 = raw[3 * 4 + k];
            tq2[0][k] = d0 - d2;
            tq2[1][k] = d1 + d2;
            tq2[2][k] = d2 - d1;
            tq2[3][k] = d1 - d3;
        };
        auto transform_rows = [&](f32x2v (&v)[16], int k) {
            v[k * 4 + 0] = tq2[k][0] - tq2[k][2];
            v[k * 4 + 1] = tq2[k][1] + tq2[k][2];
            v[k * 4 + 2] = tq2[k][2] - tq2[k][1];
            v[k * 4 + 3] = tq2[k][1] - tq2[k][3];
        };
        load_raw(0);
        load_u(u_c, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; k++) transform_cols(k);
#pragma unroll
        for (int k = 0; k < 4; k++) transform_rows(v2[0], k);
        auto load_raw1 = [&](int G, int r, int c) {
            raw[r * 4 + c] = *reinterpret_cast<const f32x2v *>(ldsb + abuf * A_BYTES + (a1[r][c] ^ (unsigned)((G >> 1) << 4)) + 8 * (G & 1));
        };
        auto load_u1 = [&](int q, int G, int s) {
            u_n[q] = *reinterpret_cast<const f32x4 *>(ldsb + B_BASE + bbuf * B_BYTES + (((G * 2 + s) * 4 + q) * 64 + lane) * 16);
        };
        static_for<0, 8>([&](auto TT) {
            constexpr int t8 = decltype(TT)::value, G = t8 >> 1, s = t8 & 1;
            // Every non-MFMA instruction is pinned into the shadow of one of the step's 16 MFMAs, a FEW per MFMA (a wave issues in order:
            // the 20 fragment reads of a step behind ONE MFMA held the next MFMA back 350-500 cycles, s_memtime per step):
            //   behind MFMAs 0..7   (even steps) the patch of the next k-group, two b64 reads each, column by column (the patch registers are
            //                       free: the previous odd step transformed both channels of the current k-group)
            //   behind MFMAs 8..11  one b128 of the next step's U; (odd steps) one column of the next k-group's input transform, 4 packed additions
            //   behind MFMAs 12..15 (odd steps) one row of it, 4 packed additions
            //   behind MFMAs 1, 6, 11 one transfer of the next stage (10 tile pieces + 8 U pieces over the 8 steps)
            static_for<0, 16>([&](auto XI) {
                constexpr int xi = decltype(XI)::value;
                if constexpr (first && t8 == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=a"(acc[xi]) : "v"(u_c[xi >> 2][xi & 3]), "v"(v2[G & 1][xi][s]));
                else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[xi]) : "v"(u_c[xi >> 2][xi & 3]), "v"(v2[G & 1][xi][s]));
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (t8 < 7) {
                    if constexpr (xi < 8 && s == 0 && G < 3) {
                        load_raw1(G + 1, (2 * xi) & 3, xi >> 1);
                        load_raw1(G + 1, (2 * xi + 1) & 3, xi >> 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (xi >= 8 && xi < 12) {
                        load_u1(xi - 8, (t8 + 1) >> 1, (t8 + 1) & 1);
                        if constexpr (s == 1) transform_cols(xi - 8);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (xi >= 12 && s == 1) {
                        transform_rows(v2[(G + 1) & 1], xi - 12);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                constexpr int q = t8 * 3 + (xi == 1 ? 0 : xi == 6 ? 1 : xi == 11 ? 2 : -100);
                if constexpr (q >= 0 && q < APW) {
                    dma_a(a_add, abuf ^ 1u, q);
                    __builtin_amdgcn_sched_barrier(0);
                } else if constexpr (q >= APW && q < APW + 8) {
                    dma_b(ob_n, sl_n, bbuf ^ 1u, q - APW);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            if constexpr (t8 < 7) {
#pragma unroll
                for (int q = 0; q < 4; q++) u_c[q] = u_n[q];
            }
        });
        W2XC_WAIT_VMCNT(0);           // the next stage's tile slice and U have landed (issued up to 8k cycles ago)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        abuf ^= 1u;
        bbuf ^= 1u;
      };
      stage(std::true_type{}, 0);
#pragma unroll 1
      for (int sl = 1; sl < nsl_rt; sl++) stage(std::false_type{}, sl);
        {
            // the hazard recogniser does not see inside inline asm: let the last MFMAs drain (16 passes) before VALU reads their results
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
            // ---- epilogue: Y = A^T M A (A^T = [1 1 1 0; 0 1 -1 -1]), bias, LeakyReLU, NHWC stores.
            //      C/D: lane&31 = block column (this lane's 2x2 block), register r = plane (r&3) + 8*(r>>2) + 4*(lane>>5) ----
            const int ob = item % NOB, pt = item / NOB;
            int tile_y, tile_x;
            tile_coords(pt, tile_y, tile_x);
            const int oy = tile_y * ROWS - d.wino_py + 4 * wave + 2 * tyl, ox = tile_x * 32 + 2 * tx;
            float *obase = d.out + (long long)oy * d.out_rs + (long long)ox * COUT + ob * 32 + 4 * kk;
            const int ty0 = tile_y * ROWS - d.wino_py;
            const bool interior = ty0 >= 0 && ty0 + ROWS <= d.out_h && tile_x * 32 + 32 <= d.out_w;   // wave-uniform
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) {
                const f32x4 bq = *reinterpret_cast<const f32x4 *>(ldsb + BIAS_BASE + (ob * 32 + 8 * q4 + 4 * kk) * 4);
                f32x4 y[2][2];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int r = 4 * q4 + e;
                    float tm[2][4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        tm[0][j] = acc[0 * 4 + j][r] + acc[1 * 4 + j][r] + acc[2 * 4 + j][r];
                        tm[1][j] = acc[1 * 4 + j][r] - acc[2 * 4 + j][r] - acc[3 * 4 + j][r];
                    }
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        const float y0 = tm[i][0] + tm[i][1] + tm[i][2] + bq[e];
                        const float y1 = tm[i][1] - tm[i][2] - tm[i][3] + bq[e];
                        // (one v_med3_f32 each: fmaxf costs extra canonicalising instructions, 17 instead of ~2 cycles beside the MFMA stream)
                        y[i][0][e] = __builtin_amdgcn_fmed3f(y0, 0.1f * y0, 3.402823466e+38f);
                        y[i][1][e] = __builtin_amdgcn_fmed3f(y1, 0.1f * y1, 3.402823466e+38f);
                    }
                }
                if (interior) {
#pragma unroll
                    for (int i = 0; i < 2; i++)
#pragma unroll
                        for (int j = 0; j < 2; j++) *reinterpret_cast<f32x4 *>(obase + (long long)i * d.out_rs + j * COUT + 8 * q4) = y[i][j];
                } else {
#pragma unroll
                    for (int i = 0; i < 2; i++)
#pragma unroll
                        for (int j = 0; j < 2; j++)
                            if (oy + i >= 0 && oy + i < d.out_h && ox + j < d.out_w)
                                *reinterpret_cast<f32x4 *>(obase + (long long)i * d.out_rs + j * COUT + 8 * q4) = y[i][j];
                }
            }
            item += per;
            if (item >= chunk_end) break;
        }
    }
    W2XC_WAIT_VMCNT(0);   // drain the speculative transfers before the LDS is released
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
bool w2xc_wino_supported(int cin, int cout)
{
    return (cin == 32 || cin == 64 || cin == 128) && (cout == 32 || cout == 64 || cout == 128);
}

size_t w2xc_wino_packed_floats(int cin, int cout) { return (size_t)16 * cin * cout; }

// wpk[plane block][slice][k-group G][step s][xi / 4][lane][xi % 4] = U_xi[o][c],  U = G g G^T  (G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1]),
// o = 32*block + (lane & 31), c = 16*slice + 8*(lane >> 5) + 2*G + s.  w is [cout][cin][3][3] (modelHandler.cpp:102); the products with
// 1/2 and 1/4 are formed in double and rounded once.
void w2xc_wino_pack(int cin, int cout, const float *w, float *dst)
{
    static const double GM[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int nsl = cin / 16, nob = cout / 32;
    for (int ob = 0; ob < nob; ob++)
        for (int sl = 0; sl < nsl; sl++)
            for (int G = 0; G < 4; G++)
                for (int s = 0; s < 2; s++)
                    for (int lane = 0; lane < 64; lane++) {
                        const int o = 32 * ob + (lane & 31), c = 16 * sl + 8 * (lane >> 5) + 2 * G + s;
                        const float *g = w + ((size_t)o * cin + c) * 9;
                        double tmp[4][3], U[4][4];
                        for (int i = 0; i < 4; i++)
                            for (int j = 0; j < 3; j++) tmp[i][j] = GM[i][0] * g[0 * 3 + j] + GM[i][1] * g[1 * 3 + j] + GM[i][2] * g[2 * 3 + j];
                        for (int i = 0; i < 4; i++)
                            for (int j = 0; j < 4; j++) U[i][j] = tmp[i][0] * GM[j][0] + tmp[i][1] * GM[j][1] + tmp[i][2] * GM[j][2];
                        for (int xi = 0; xi < 16; xi++)
                            dst[((((((size_t)ob * nsl + sl) * 4 + G) * 2 + s) * 4 + (xi >> 2)) * 64 + lane) * 4 + (xi & 3)] = (float)U[xi >> 2][xi & 3];
                    }
}

template <int CIN, int COUT>
static hipError_t launch_wino(const W2xcConvDesc &d, hipStream_t stream)
{
    const int tiles_x = (d.out_w + 31) / 32, tiles_y = (d.out_h + (d.wino_py & 1) + 15) / 16;
    const int nitems = tiles_x * tiles_y * (COUT / 32);
    constexpr size_t lds_bytes = 2 * (size_t)(4 * 10 * 1024) + 2 * (size_t)(32 * 1024) + 10 * 1024 + COUT * 4;   // tile + U ring + the DMA offset table + bias
    static_assert(lds_bytes <= 160 * 1024, "LDS budget");
    auto kern = conv3x3_wino<CIN, COUT>;
    static std::atomic<unsigned long long> attr_done{0};   // function attributes are per device
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 64 || !((attr_done.load() >> dev) & 1ull)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        if (dev < 64) attr_done.fetch_or(1ull << dev);
    }
    int grid = 256;   // one persistent workgroup per CU; a multiple of 8 (one share per XCD)
    if (grid > ((nitems + 7) & ~7)) grid = (nitems + 7) & ~7;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, stream, d, tiles_x, nitems);
    return hipGetLastError();
}

// d.wpk = w2xc_wino_pack image; NHWC fp32 in / out like W2XC_K_MFMA
hipError_t w2xc_launch_wino(const W2xcConvDesc &d, hipStream_t stream)
{
    if (d.out_w <= 0 || d.out_h <= 0) return hipSuccess;
    if (d.in_ps != d.cin || d.in_cs != 1 || d.out_ps != d.cout || d.out_cs != 1 || d.in_shift != 0) return hipErrorInvalidValue;
    if ((d.in_rs & 3) != 0 || (d.out_rs & 3) != 0) return hipErrorInvalidValue;   // 16-byte accesses
    switch (d.cin * 1000 + d.cout) {
    case 32032:  return launch_wino<32, 32>(d, stream);
    case 32064:  return launch_wino<32, 64>(d, stream);
    case 32128:  return launch_wino<32, 128>(d, stream);
    case 64032:  return launch_wino<64, 32>(d, stream);
    case 128032: return launch_wino<128, 32>(d, stream);
    case 64064:  return launch_wino<64, 64>(d, stream);
    case 64128:  return launch_wino<64, 128>(d, stream);
    case 128064: return launch_wino<128, 64>(d, stream);
    case 128128: return launch_wino<128, 128>(d, stream);
    default: return hipErrorInvalidValue;
    }
}
