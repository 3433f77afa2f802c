// w2xc_wino16.hip -- conv3x3_wino16: the 3x3 x Cin x Cout contraction of Model::filterWorker
// (/root/reference/src/modelHandler.cpp:117-159) as Winograd F(2x2, 3x3) on v_mfma_f32_16x16x4_f32, two workgroups per CU.
//
// Same algebra as conv3x3_wino (w2xc_wino.hip):  Y = A^T [ (G g G^T) (.) (B^T d B) ] A  per 2x2 output block, the 16 positions xi of
// the transformed domain being 16 independent GEMMs  M_xi[o][t] = sum_c U_xi[o][c] V_xi[c][t]  (o = output plane, t = block,
// c = input plane), fp32 throughout, V transformed in registers by the lane that owns the block.  What is different is the shape:
//
//   conv3x3_wino   32x32x2 MFMA, a wave owns 32 planes x 32 blocks x 16 xi = 256 accumulators = the whole register file of a
//                  one-wave-per-SIMD kernel.  Whatever the wave issues besides MFMAs (patch reads, the transform's additions, the
//                  transfers, the whole epilogue) is time the matrix pipe of that SIMD idles: measured 0.55-0.72 of the MFMA peak.
//   conv3x3_wino16 16x16x4 MFMA (same rate, 32 cycles): a wave owns 32 planes x 16 blocks x 16 xi = 128 accumulators, so TWO waves
//                  fit a SIMD -- and they come from two INDEPENDENT 4-wave workgroups (<= 80 KiB of LDS each), which drift apart
//                  by construction: one workgroup's epilogue, barrier wait or stage head runs under the other one's MFMAs.
//
//   Work item  8 rows x 32 pixels of output (4 x 16 blocks) x one block of 32 output planes; wave w owns block row w.
//              Lane (t = lane & 15, k = lane >> 4): block t of the row; K index k of the MFMA = channels 2k, 2k+1 of a slice.
//   Stage      one 8-channel slice: 2 steps (channel 2k + st in lane quarter k) x 16 xi x 2 plane tiles = 64 MFMAs = 2048 cycles.
//              A lane reads its 4x4 patch once per stage (16 ds_read_b64 = both channels), transforms both channels with packed
//              additions, and every V value feeds two MFMAs (plane tiles 0 and 1).  The patch of stage s+1 is read and transformed
//              under the MFMAs of stage s.
//   LDS        A[3] x 12 KiB: the 10 x 34 pixel halo tile of a slice (32 B per pixel) by LDS-DMA, three stages deep, laid out
//              [row][16-byte chunk 0/1][column parity][17 columns]: the 16 lanes of a quarter read 16 CONSECUTIVE 16-byte
//              entries and the two quarters of a ds_read_b64 lane group take the two halves of them -- conflict-free without
//              a swizzle, and every patch element is ONE base register + an immediate offset.
//              U[2] x 16 KiB: the weights of (plane block, slice) in fragment order [step][plane tile][xi/4][lane][xi%4].
//              + 3 KiB per-lane transfer offsets + bias.  72 KiB per workgroup.
//   Transfers  SGPR base + 32-bit lane offset (no 64-bit VALU address per piece); per wave and stage 4 U pieces (one stage ahead)
//              then 3 tile pieces (three stages ahead); the stage closes with a COUNTED vmcnt that leaves the tile pieces in flight.
//   Epilogue   output transform, bias, LeakyReLU, 16-byte NHWC stores (a lane holds 4 consecutive planes of its block's 4 pixels).
//   Banding    blocks sit on EVEN rows of the layer's whole output (W2xcConvDesc::wino_py), as in conv3x3_wino.
#include "w2xc_kernels.h"
#include "w2xc_device.h"

#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <type_traits>

typedef float f32x2v __attribute__((ext_vector_type(2)));

// ABL: timing-only ablations (wrong results; -DW16_ABLATE builds): 1 = no U transfers in the stage loop, 2 = no tile transfers,
// 4 = no epilogue stores, 8 = no stage barrier
template <int CIN, int COUT, int ABL = 0>
__global__ void __launch_bounds__(256, 2) conv3x3_wino16(W2xcConvDesc d, int tiles_x, int nitems)
{
    constexpr int ROWS = 8, HW = 34, HH = ROWS + 2;
    constexpr int NSL = CIN / 8, NOB = COUT / 32, NW = 4;
    constexpr int ROW_SLOTS = 4 * 17;                       // 16-byte slots of one halo row: [chunk 2][parity 2][17]
    constexpr int A_SLOTS = HH * ROW_SLOTS;                 // 680
    constexpr int APW = (A_SLOTS + NW * 64 - 1) / (NW * 64);   // 3 pieces of 1 KiB per wave
    constexpr int A_PIECES = APW * NW;                      // 12 (the last one only re-reads the tile's last slot: no wave-dependent branch)
    constexpr unsigned A_BYTES = A_PIECES * 1024;
    constexpr unsigned U_BYTES = 16 * 1024;
    constexpr unsigned U_BASE = 3 * A_BYTES;
    constexpr unsigned LOFS_BASE = U_BASE + 2 * U_BYTES;
    constexpr unsigned BIAS_BASE = LOFS_BASE + APW * 256 * 4;
    static_assert(CIN % 8 == 0 && NSL >= 4 && COUT % 32 == 0, "planes");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    const char *ldsb = reinterpret_cast<const char *>(lds);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = lane & 15, k = lane >> 4;

    // persistent schedule: XCD x (= blockIdx % 8) walks its own contiguous chunk of the item list
    const int xcd = blockIdx.x & 7, per = gridDim.x >> 3;
    const int cq = nitems >> 3, cr = nitems & 7;
    const int chunk_begin = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
    const int chunk_end = chunk_begin + cq + (xcd < cr ? 1 : 0);
    int item = chunk_begin + (blockIdx.x >> 3);
    if (item >= chunk_end) return;

    // ---- tile transfers: slot s of a slice <- pixel (row, col), chunk c4 ----
    auto slot_of = [&](int jj, int &row, int &col, int &c4) {
        int s = (jj * NW + wave) * 64 + lane;
        s = s < A_SLOTS ? s : A_SLOTS - 1;                  // slots past the tile re-read its last one
        row = s / ROW_SLOTS;
        const int rem = s - row * ROW_SLOTS;
        c4 = rem / 34;
        const int rem2 = rem - c4 * 34;
        const int par = rem2 / 17;
        col = 2 * (rem2 - par * 17) + par;
    };
    for (int c = threadIdx.x; c < COUT; c += 256) lds[BIAS_BASE / 4 + c] = d.bias[c];   // (visible after the prologue barrier)
    {
        unsigned *lofs = reinterpret_cast<unsigned *>(const_cast<char *>(ldsb) + LOFS_BASE);
#pragma unroll
        for (int jj = 0; jj < APW; jj++) {
            int row, col, c4;
            slot_of(jj, row, col, c4);
            lofs[jj * 256 + threadIdx.x] = (unsigned)(((long long)row * d.in_rs + (long long)col * CIN + 4 * c4) * 4);   // bytes
        }
    }
    // The transfer source of (item, slice) = wave-uniform 64-bit base (tile origin, clamped into the plane) + voff[jj] (32-bit, per lane)
    unsigned voff[APW];
    const char *a_base;
    auto tile_offsets = [&](int it) {
        const int pt = it / NOB;
        const int ty_ = pt / tiles_x, tx_ = pt - ty_ * tiles_x;
        const int y0 = ty_ * ROWS - d.wino_py + d.off_y, x0 = tx_ * 32 + d.off_x;
        const int yb = clampi(y0, 0, d.in_h - 1), xb = clampi(x0, 0, d.in_w - 1);
        a_base = reinterpret_cast<const char *>(d.in) + ((long long)yb * d.in_rs + (long long)xb * CIN) * 4;
        if (y0 >= 0 && y0 + HH <= d.in_h && x0 >= 0 && x0 + HW <= d.in_w) {   // wave-uniform
            const unsigned *lofs = reinterpret_cast<const unsigned *>(ldsb + LOFS_BASE);
#pragma unroll
            for (int jj = 0; jj < APW; jj++) voff[jj] = lofs[jj * 256 + threadIdx.x];
            return;
        }
#pragma unroll
        for (int jj = 0; jj < APW; jj++) {
            int row, col, c4;
            slot_of(jj, row, col, c4);
            const int gy = clampi(y0 + row, 0, d.in_h - 1) - yb;
            const int gx = clampi(x0 + col, 0, d.in_w - 1) - xb;
            voff[jj] = (unsigned)(((long long)gy * d.in_rs + (long long)gx * CIN + 4 * c4) * 4);
        }
    };
    // tile piece jj of slice sl_ -> tile slot `slot`
    auto dma_a = [&](int sl_, unsigned slot, int jj) {
        const char *sbase = a_base + sl_ * 32;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(lds0 + slot * A_BYTES + (unsigned)(jj * NW + wave) * 1024u);
        lds_dma16_s<0>(sbase, voff[jj], dst);
    };
    // U of (plane block ob, slice sl_): 16 pieces of 1 KiB, 4 per wave
    const unsigned b_voff = (unsigned)lane * 16u;
    auto dma_b = [&](int ob, int sl_, unsigned slot, int jb) {
        const char *sbase = reinterpret_cast<const char *>(d.wpk) + ((size_t)(ob * NSL + sl_) * 16 + wave * 4) * 1024;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(lds0 + U_BASE + slot * U_BYTES + (unsigned)(wave * 4) * 1024u);
        switch (jb) {
        case 0: lds_dma16_s<0>(sbase, b_voff, dst); break;
        case 1: lds_dma16_s<1024>(sbase, b_voff, dst); break;
        case 2: lds_dma16_s<2048>(sbase, b_voff, dst); break;
        default: lds_dma16_s<3072>(sbase, b_voff, dst); break;
        }
    };

    // ---- fragment addressing ----
    // patch element (r, c) of this lane's block: halo pixel (2w + r, 2t + c), channels 2k, 2k+1 of the slice = chunk k >> 1, half k & 1:
    //   byte ((2w + r) * 68 + (k >> 1) * 34 + (c & 1) * 17 + t + (c >> 1)) * 16 + 8 * (k & 1)   -- base + immediate
    const unsigned pbase = (unsigned)(((2 * wave) * ROW_SLOTS + (k >> 1) * 34 + t) * 16 + 8 * (k & 1));
    constexpr unsigned P_ROW = ROW_SLOTS * 16, P_PAR = 17 * 16;
    const unsigned ubase = U_BASE + (unsigned)lane * 16u;

    f32x2v va[16], vb[16]; // V of the stage being multiplied / of the next one (roles alternate): [xi] = (channel 2k, channel 2k+1)
    f32x2v raw[16], tq[4][4];
    // B^T d B with B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1], columns of d first
    auto transform_cols = [&](int c) {
        const f32x2v d0 = raw[0 * 4 + c], d1 = raw[1 * 4 + c], d2 = raw[2 * 4 + c], d3 = raw[3 * 4 + c];
        tq[0][c] = d0 - d2;
        tq[1][c] = d1 + d2;
        tq[2][c] = d2 - d1;
        tq[3][c] = d1 - d3;
        // (pinned: the results are first used a stage later, and LLVM sinks the additions there -- out of the MFMA shadows -- otherwise)
        asm volatile("" : "+v"(tq[0][c]), "+v"(tq[1][c]), "+v"(tq[2][c]), "+v"(tq[3][c]));
    };
    auto transform_rows = [&](f32x2v (&v)[16], int r) {
        v[r * 4 + 0] = tq[r][0] - tq[r][2];
        v[r * 4 + 1] = tq[r][1] + tq[r][2];
        v[r * 4 + 2] = tq[r][2] - tq[r][1];
        v[r * 4 + 3] = tq[r][1] - tq[r][3];
        asm volatile("" : "+v"(v[r * 4 + 0]), "+v"(v[r * 4 + 1]), "+v"(v[r * 4 + 2]), "+v"(v[r * 4 + 3]));
    };

    // ---- prologue: tile slices 0..2 and U(slice 0) of the first item; V of its first stage ----
    tile_offsets(item);
#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
        for (int jj = 0; jj < APW; jj++) dma_a(s, (unsigned)s, jj);
#pragma unroll
    for (int jb = 0; jb < 4; jb++) dma_b(item % NOB, 0, 0, jb);
    W2XC_WAIT_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++)
            raw[r * 4 + c] = *reinterpret_cast<const f32x2v *>(ldsb + pbase + r * P_ROW + (c & 1) * P_PAR + (c >> 1) * 16);
#pragma unroll
    for (int c = 0; c < 4; c++) transform_cols(c);
#pragma unroll
    for (int r = 0; r < 4; r++) transform_rows(va, r);

    unsigned aslot = 0;    // tile slot of the CURRENT stage's slice (its patch was read one stage ago); runs 0, 1, 2, 0, ... over all stages
    unsigned uslot = 0;
    for (;;) {
        const int item_n = item + per < chunk_end ? item + per : item;   // (the last item prefetches itself: harmless)
        // The accumulators are DEFINED by the first stage of an item (C = 0) and die in its epilogue
        f32x4 acc[16][2];
        auto stage = [&](auto FIRST, int sl, f32x2v (&vcur)[16], f32x2v (&vnext)[16]) {
            constexpr bool first = decltype(FIRST)::value;
            // transfers of this stage: U of the next stage, tile slice of the stage three ahead
            int u_ob = item % NOB, u_sl = sl + 1;
            if (sl == NSL - 1) { u_ob = item_n % NOB; u_sl = 0; }
            int a_sl = sl + 3;
            if (sl == NSL - 3) tile_offsets(item_n);          // from here on the tile prefetch runs in the next item
            if (sl >= NSL - 3) a_sl = sl + 3 - NSL;
            const unsigned a_dst = aslot;                      // slice (stage + 3) replaces the slice whose patch was read a stage ago
            const unsigned a_src = aslot == 2 ? 0u : aslot + 1u;   // slice (stage + 1): transformed during this stage
            const char *pa = ldsb + a_src * A_BYTES + pbase;
            const char *ua = ldsb + ubase + uslot * U_BYTES;
            f32x4 u[2][2];
            u[0][0] = *reinterpret_cast<const f32x4 *>(ua + 0 * 1024);
            u[0][1] = *reinterpret_cast<const f32x4 *>(ua + 4 * 1024);
            static_for<0, 64>([&](auto SLOT) {
                constexpr int slot = decltype(SLOT)::value;
                constexpr int st = slot >> 5, g = slot >> 3, x4 = (slot >> 1) & 3, pt = slot & 1, xi = ((slot >> 3) & 3) * 4 + x4;
                if constexpr (first && st == 0) {
                    const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
                    acc[xi][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[g & 1][pt][x4], vcur[xi][st], z, 0, 0, 0);
                } else {
                    acc[xi][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[g & 1][pt][x4], vcur[xi][st], acc[xi][pt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- fillers, a few per MFMA ----
                // U fragments of the next group of 8 MFMAs (group g = (step, xi/4); [step][plane tile][xi/4] KiB in the U slot)
                if constexpr ((slot & 7) == 1 && g < 7) {
                    constexpr int gn = g + 1;
                    u[gn & 1][0] = *reinterpret_cast<const f32x4 *>(ua + (((gn >> 2) * 2 + 0) * 4 + (gn & 3)) * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr ((slot & 7) == 5 && g < 7) {
                    constexpr int gn = g + 1;
                    u[gn & 1][1] = *reinterpret_cast<const f32x4 *>(ua + (((gn >> 2) * 2 + 1) * 4 + (gn & 3)) * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the next stage's patch, column by column (slots 0..15), its transform behind it (columns 8..23, rows 24..39)
                if constexpr (slot < 16) {
                    constexpr int r = slot & 3, c = slot >> 2;
                    raw[r * 4 + c] = *reinterpret_cast<const f32x2v *>(pa + r * P_ROW + (c & 1) * P_PAR + (c >> 1) * 16);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (slot >= 8 && slot < 24 && (slot & 3) == 3) {
                    transform_cols((slot - 8) >> 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (slot >= 24 && slot < 40 && (slot & 3) == 3) {
                    transform_rows(vnext, (slot - 24) >> 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // transfers: U pieces first, tile pieces last (the stage's closing wait leaves the tile pieces in flight)
                if constexpr (slot == 18 || slot == 26 || slot == 34 || slot == 42) {
                    if constexpr (!(ABL & 1)) dma_b(u_ob, u_sl, uslot ^ 1u, (slot - 18) >> 3);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (slot == 46 || slot == 52 || slot == 58) {
                    if constexpr (!(ABL & 2)) dma_a(a_sl, a_dst, (slot - 46) / 6);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            // U(next stage) and the tile slice two stages ahead have landed; this stage's tile pieces (the youngest) may still fly
            W2XC_WAIT_VMCNT(APW);
            if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            aslot = a_src;
            uslot ^= 1u;
        };
        stage(std::true_type{}, 0, va, vb);
#pragma unroll 1
        for (int sl = 1; sl < NSL - 1; sl += 2) {
            stage(std::false_type{}, sl, vb, va);
            stage(std::false_type{}, sl + 1, va, vb);
        }
        stage(std::false_type{}, NSL - 1, vb, va);
        {
            // ---- epilogue: Y = A^T M A (A^T = [1 1 1 0; 0 1 -1 -1]), bias, LeakyReLU, NHWC stores.
            //      C/D of the 16x16 MFMA: lane & 15 = block, register e = plane 4 * (lane >> 4) + e of the plane tile ----
            const int ob = item % NOB, ptile = item / NOB;
            const int tile_y = ptile / tiles_x, tile_x = ptile - tile_y * tiles_x;
            const int ty0 = tile_y * ROWS - d.wino_py;
            const int oy = ty0 + 2 * wave, ox = tile_x * 32 + 2 * t;
            float *obase = d.out + (long long)oy * d.out_rs + (long long)ox * COUT + ob * 32 + 4 * k;
            const bool interior = ty0 >= 0 && ty0 + ROWS <= d.out_h && tile_x * 32 + 32 <= d.out_w;   // wave-uniform
#pragma unroll
            for (int pt = 0; pt < 2; pt++) {
                const f32x4 bq = *reinterpret_cast<const f32x4 *>(ldsb + BIAS_BASE + (ob * 32 + 16 * pt + 4 * k) * 4);
                f32x4 y[2][2];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float tm[2][4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        tm[0][j] = acc[0 * 4 + j][pt][e] + acc[1 * 4 + j][pt][e] + acc[2 * 4 + j][pt][e];
                        tm[1][j] = acc[1 * 4 + j][pt][e] - acc[2 * 4 + j][pt][e] - acc[3 * 4 + j][pt][e];
                    }
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        const float y0 = tm[i][0] + tm[i][1] + tm[i][2] + bq[e];
                        const float y1 = tm[i][1] - tm[i][2] - tm[i][3] + bq[e];
                        y[i][0][e] = fmaxf(y0, 0.1f * y0);
                        y[i][1][e] = fmaxf(y1, 0.1f * y1);
                    }
                }
                if constexpr ((ABL & 4) != 0) {
                    if (y[0][0][0] == 12345.678f) *reinterpret_cast<f32x4 *>(obase) = y[0][0] + y[0][1] + y[1][0] + y[1][1];
                } else if (interior) {
#pragma unroll
                    for (int i = 0; i < 2; i++)
#pragma unroll
                        for (int j = 0; j < 2; j++) *reinterpret_cast<f32x4 *>(obase + (long long)i * d.out_rs + j * COUT + 16 * pt) = y[i][j];
                } else {
#pragma unroll
                    for (int i = 0; i < 2; i++)
#pragma unroll
                        for (int j = 0; j < 2; j++)
                            if (oy + i >= 0 && oy + i < d.out_h && ox + j < d.out_w)
                                *reinterpret_cast<f32x4 *>(obase + (long long)i * d.out_rs + j * COUT + 16 * pt) = y[i][j];
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
bool w2xc_wino16_supported(int cin, int cout)
{
    return (cin == 32 || cin == 64 || cin == 128) && (cout == 32 || cout == 64 || cout == 128);
}

// wpk[plane block][slice (8 channels)][step st][plane tile pt][xi / 4][lane][xi % 4] = U_xi[o][c],  U = G g G^T
// (G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1]),  o = 32*block + 16*pt + (lane & 15),  c = 8*slice + 2*(lane >> 4) + st.
// w is [cout][cin][3][3] (modelHandler.cpp:102); the products with 1/2 and 1/4 are formed in double and rounded once.
void w2xc_wino16_pack(int cin, int cout, const float *w, float *dst)
{
    static const double GM[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int nsl = cin / 8, nob = cout / 32;
    for (int ob = 0; ob < nob; ob++)
        for (int sl = 0; sl < nsl; sl++)
            for (int st = 0; st < 2; st++)
                for (int pt = 0; pt < 2; pt++)
                    for (int lane = 0; lane < 64; lane++) {
                        const int o = 32 * ob + 16 * pt + (lane & 15), c = 8 * sl + 2 * (lane >> 4) + st;
                        const float *g = w + ((size_t)o * cin + c) * 9;
                        double tmp[4][3], U[4][4];
                        for (int i = 0; i < 4; i++)
                            for (int j = 0; j < 3; j++) tmp[i][j] = GM[i][0] * g[0 * 3 + j] + GM[i][1] * g[1 * 3 + j] + GM[i][2] * g[2 * 3 + j];
                        for (int i = 0; i < 4; i++)
                            for (int j = 0; j < 4; j++) U[i][j] = tmp[i][0] * GM[j][0] + tmp[i][1] * GM[j][1] + tmp[i][2] * GM[j][2];
                        for (int xi = 0; xi < 16; xi++)
                            dst[((((((size_t)ob * nsl + sl) * 2 + st) * 2 + pt) * 4 + (xi >> 2)) * 64 + lane) * 4 + (xi & 3)] = (float)U[xi >> 2][xi & 3];
                    }
}

template <int CIN, int COUT, int ABL = 0>
static hipError_t launch_wino16(const W2xcConvDesc &d, hipStream_t stream)
{
    const int tiles_x = (d.out_w + 31) / 32, tiles_y = (d.out_h + (d.wino_py & 1) + 7) / 8;
    const int nitems = tiles_x * tiles_y * (COUT / 32);
    constexpr size_t lds_bytes = 3 * (size_t)(12 * 1024) + 2 * (size_t)(16 * 1024) + 3 * 1024 + COUT * 4;   // tile ring + U ring + offset table + bias
    static_assert(2 * lds_bytes <= 160 * 1024, "two workgroups per CU");
    auto kern = conv3x3_wino16<CIN, COUT, ABL>;
    static std::atomic<unsigned long long> attr_done{0};   // function attributes are per device
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 64 || !((attr_done.load() >> dev) & 1ull)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        if (dev < 64) attr_done.fetch_or(1ull << dev);
    }
    int grid = 512;   // two persistent workgroups per CU; a multiple of 8 (one share per XCD)
    if (grid > ((nitems + 7) & ~7)) grid = (nitems + 7) & ~7;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, stream, d, tiles_x, nitems);
    return hipGetLastError();
}

// d.wpk = w2xc_wino16_pack image; NHWC fp32 in / out like W2XC_K_MFMA
hipError_t w2xc_launch_wino16(const W2xcConvDesc &d, hipStream_t stream)
{
    if (d.out_w <= 0 || d.out_h <= 0) return hipSuccess;
    if (d.in_ps != d.cin || d.in_cs != 1 || d.out_ps != d.cout || d.out_cs != 1 || d.in_shift != 0) return hipErrorInvalidValue;
    if ((d.in_rs & 3) != 0 || (d.out_rs & 3) != 0) return hipErrorInvalidValue;   // 16-byte accesses
#ifdef W16_ABLATE
    static const int abl = [] { const char *e = getenv("W2XC_W16_ABL"); return e ? atoi(e) : 0; }();
    if (d.cin == 128 && d.cout == 128) switch (abl) {
        case 1: return launch_wino16<128, 128, 1>(d, stream);
        case 2: return launch_wino16<128, 128, 2>(d, stream);
        case 3: return launch_wino16<128, 128, 3>(d, stream);
        case 4: return launch_wino16<128, 128, 4>(d, stream);
        case 7: return launch_wino16<128, 128, 7>(d, stream);
        case 15: return launch_wino16<128, 128, 15>(d, stream);
        default: break;
    }
    if (d.cin == 32 && d.cout == 32) switch (abl) {
        case 3: return launch_wino16<32, 32, 3>(d, stream);
        case 4: return launch_wino16<32, 32, 4>(d, stream);
        case 7: return launch_wino16<32, 32, 7>(d, stream);
        default: break;
    }
#endif
    switch (d.cin * 1000 + d.cout) {
    case 32032:  return launch_wino16<32, 32>(d, stream);
    case 32064:  return launch_wino16<32, 64>(d, stream);
    case 32128:  return launch_wino16<32, 128>(d, stream);
    case 64032:  return launch_wino16<64, 32>(d, stream);
    case 128032: return launch_wino16<128, 32>(d, stream);
    case 64064:  return launch_wino16<64, 64>(d, stream);
    case 64128:  return launch_wino16<64, 128>(d, stream);
    case 128064: return launch_wino16<128, 64>(d, stream);
    case 128128: return launch_wino16<128, 128>(d, stream);
    default: return hipErrorInvalidValue;
    }
}
