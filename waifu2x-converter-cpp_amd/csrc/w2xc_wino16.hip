// w2xc_wino16.hip -- conv3x3_wino16: the 3x3 x Cin x Cout contraction of Model::filterWorker
// (/root/reference/src/modelHandler.cpp:117-159) as Winograd F(2x2, 3x3) on v_mfma_f32_16x16x4_f32, two waves per SIMD.
//
// Same algebra as conv3x3_wino (w2xc_wino.hip):  Y = A^T [ (G g G^T) (.) (B^T d B) ] A  per 2x2 output block, the 16 positions xi of
// the transformed domain being 16 independent GEMMs  M_xi[o][t] = sum_c U_xi[o][c] V_xi[c][t]  (o = output plane, t = block,
// c = input plane), fp32 throughout, V transformed in registers by the lane that owns the block.  What is different is the shape:
//
//   conv3x3_wino   32x32x2 MFMA, a wave owns 32 planes x 32 blocks x 16 xi = 256 accumulators = the whole register file of a
//                  one-wave-per-SIMD kernel.  Whatever the wave issues besides MFMAs (patch reads, the transform's additions, the
//                  transfers, the whole epilogue) is time the matrix pipe of that SIMD idles: measured 0.55-0.72 of the MFMA peak.
//   conv3x3_wino16 16x16x4 MFMA (same rate, 32 cycles): a wave owns 32 planes x 16 blocks x 16 xi = 128 accumulators, so TWO waves
//                  fit a SIMD.  The workgroup is 8 waves = 2 PLANE GROUPS x 4 block rows: both groups work on the same pixel tile
//                  (one tile transfer serves 64 output planes), in the same stage.  (A variant with group 1 one stage behind group 0 --
//                  so that one group's epilogue runs under the other's MFMAs -- measured the same and was dropped: the fp32 MFMA
//                  and the VALU share the fp32 lanes, an epilogue costs its ALU cycles wherever it runs.)
//
//   Work item  8 rows x 32 pixels of output (4 x 16 blocks) x 64 output planes; wave (g, w): plane group g, block row w.
//              Lane (t = lane & 15, k = lane >> 4): block t of the row; K index k of the MFMA = channels 2k, 2k+1 of a slice.
//   Stage      one 8-channel slice: 2 steps (channel 2k + st in lane quarter k) x 16 xi x 2 plane tiles = 64 MFMAs = 2048 cycles
//              per wave.  A lane reads its 4x4 patch once per stage (16 ds_read_b64 = both channels), transforms both channels with
//              64 SCALAR additions in four bunches of 16 (beside the fp32 MFMA a VALU instruction costs 2 matrix-pipe cycles + ~4.5 per
//              MFMA gap that holds VALU at all, packing buys nothing: tools/ubench/mfma_fillers.hip), and every V value feeds two
//              MFMAs (plane tiles 0 and 1).  The patch of stage s+1 is read and transformed under the MFMAs of stage s.
//   LDS        A[3] x 27 KiB: the 10 x 34 pixel halo tile of a 16-CHANNEL slice (two stages), pixel-major: a pixel's four 16-byte
//              chunks + one pad slot = 80 B, pixels of a row even columns first, then odd ones.  The 80-byte stride makes the 16
//              lanes of a quarter (blocks 2 pixels apart = consecutive positions) hit 16 different bank quads (5 t mod 16), the
//              quarter pairs of a ds_read_b64 lane group take the two halves: conflict-free, ONE base register + immediates.
//              And a transfer instruction's 64 lanes cover 13 pixels x 64 contiguous bytes (the first version's channel-major
//              layout touched 64 different cache lines per instruction: 1.9 of its 10.7 ms on 128->128).
//              U[2 groups][2] x 16 KiB: the weights of (plane block, slice) in fragment order [step][plane tile][xi/4][lane][xi%4].
//              + 8 KiB per-lane transfer offsets + 1 KiB dump + bias: 155 KiB, one workgroup per CU.
//   Transfers  SGPR base + 32-bit lane offset; per wave and stage 4 U pieces (own group, one stage ahead), and every second stage 4 tile
//              pieces (two 16-channel slices ahead); U first, tile pieces last, so the stage's closing COUNTED vmcnt leaves the
//              tile pieces in flight.
//   Epilogue   output transform on register pairs, bias, LeakyReLU (one v_med3), 16-byte NHWC stores (a lane holds 4 consecutive planes of
//              its block's 4 pixels), at raised wave priority; FUSE: the one-plane last layer on the fresh activations (below).
//   Tile walk  strips of 16 tiles, row by row inside a strip: an XCD's next round is the tile row below, its halo rows still in L2.
//   Banding    blocks sit on EVEN rows of the layer's whole output (W2xcConvDesc::wino_py), as in conv3x3_wino.
// Measured (round 3, 2160x3840, profiles/r3_*): 128->128 8.7-8.9 ms (conv3x3_wino 9.6, conv3x3_mfma2 16.9) = 0.78-0.80 of the fp32 MFMA peak on the
// multiplies it issues, SQ_LDS_BANK_CONFLICT ~ 0, HBM traffic 1.03-1.18x algorithmic, no register spills; what the rest is: DESIGN.md 3.
#include "w2xc_kernels.h"
#include "w2xc_device.h"

#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <type_traits>

typedef float f32x2v __attribute__((ext_vector_type(2)));

#ifdef W16_TIMING
// tools/ubench/wino16_timing.hip: s_memtime stamps of block row 0 of each plane group of workgroup 0 -- per stage (before the closing
// wait, after it, after the barrier) and after every epilogue -- [group][index].  (The stamps are stores: they perturb what they measure.)
__device__ unsigned long long w16_stamps[2][4096];
#define W16_STAMP(idx) do { const int i_ = (idx); if (blockIdx.x == 0 && brow == 0 && lane == 0 && i_ < 4096) w16_stamps[grp][i_] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W16_STAMP(idx) do { } while (0)
#endif

// ABL: timing-only ablations (wrong results; -DW16_ABLATE builds, W2XC_W16_ABL=<bits> picks one at run time, profiles/r3_sweeps.log has the numbers):
//   1 no U transfers in the stage loop | 2 no tile transfers | 4 no epilogue stores | 8 no stage barrier | 16 no patch reads / input transform |
//   32 no U fragment reads | 64 patch reads but no transform additions | 512 tile transfers always read the first tile (L2 hits) |
//   1024 tile pieces early in the stage | 2048 epilogue at normal priority | 256 transform additions two per MFMA gap instead of four bunches of 16 |
//   128 a correct VARIANT: accumulators forced into AGPRs (inline-asm MFMA)
// FUSE: the NEXT layer is the model's last one with ONE output plane (every waifu2x model): it is computed in this kernel's epilogue
// ("taps as rows": G[tap][pixel] = sum_c W_last[c][tap] act[c][pixel], 32 more MFMAs per item and wave, on the activations the epilogue
// has just produced) and `out` receives the partial tap planes G[32-plane block][tap][y][x] (strides d.out_ts / out_gs / out_rs / out_ps) that
// conv3x3_last_gather sums -- 144 instead of 512 bytes per pixel written, and the 512 the last layer would read never exist.
template <int CIN, int COUT, int ABL = 0, bool FUSE = false>
__global__ void __launch_bounds__(512, 2) conv3x3_wino16(W2xcConvDesc d, int tiles_x, int nitems)
{
    constexpr int ROWS = 8, HW = 34, HH = ROWS + 2;
    constexpr int NSL = CIN / 8;                            // stages (8-channel slices) per item
    constexpr int NSP = CIN / 16;                           // 16-channel tile slices per item
    constexpr int NOB = COUT / 64;                          // 64-plane blocks
    constexpr int NW = 8;
    constexpr int PIX = 5;                                  // 16-byte slots per pixel: 4 chunks + 1 pad (80-byte stride)
    constexpr int A_SLOTS = HH * HW * PIX;                  // 1700
    constexpr int A_PIECES = (A_SLOTS + 63) / 64;           // 27 pieces of 1 KiB
    constexpr int APW = (A_PIECES + NW - 1) / NW;           // 4 per wave; pieces >= A_PIECES land in the dump KiB
    constexpr unsigned A_BYTES = A_PIECES * 1024;
    constexpr unsigned DUMP_BASE = 3 * A_BYTES;
    constexpr unsigned U_BASE = DUMP_BASE + 1024;
    constexpr unsigned U_BYTES = 16 * 1024;
    constexpr unsigned LOFS_BASE = U_BASE + 4 * U_BYTES;
    constexpr unsigned BIAS_BASE = LOFS_BASE + APW * 512 * 4;
    static_assert(CIN % 16 == 0 && NSP >= 2 && COUT % 64 == 0, "planes");
    // Pixel tiles are walked in STRIPS of 16 tiles (512 pixels) across, row by row inside a strip: the 32 CUs of an XCD work on 16 neighbouring
    // tiles x 2 items at a time, so the next round of an XCD is the tile row BELOW -- whose two halo rows are then still in that XCD's L2
    // (a plain row-major walk re-reads them from HBM a whole plane row later: 1.33x the plane, measured).
    constexpr int STRIP = 16;
    const int tiles_y = nitems / (NOB * tiles_x);
    auto tile_coords = [&](int pt, int &ty_, int &tx_) {
        const int per_strip = STRIP * tiles_y;
        int sidx = pt / per_strip;
        const int nfull = tiles_x / STRIP;
        if (sidx > nfull) sidx = nfull;                          // (the last, narrower strip)
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
    const int grp = wave >> 2, brow = wave & 3;
    const int t = lane & 15, k = lane >> 4;

    // persistent schedule: XCD x (= blockIdx % 8) walks its own contiguous chunk of the item list; this workgroup's items are
    // item_of(0), item_of(1), ...
    const int xcd = blockIdx.x & 7, per = gridDim.x >> 3;
    const int cq = nitems >> 3, cr = nitems & 7;
    const int chunk_begin = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
    const int chunk_end = chunk_begin + cq + (xcd < cr ? 1 : 0);
    const int item0 = chunk_begin + (blockIdx.x >> 3);
    if (item0 >= chunk_end) return;
    const int nmy = (chunk_end - item0 + per - 1) / per;
    auto item_of = [&](int n) { return item0 + (n < nmy ? n : nmy - 1) * per; };   // (past the end: the last item again, harmless)

    // ---- tile transfers: slot s of a 16-channel slice <- pixel (row, col), chunk ch ----
    auto slot_of = [&](int jj, int &row, int &col, int &ch) {
        int s = (jj * NW + wave) * 64 + lane;
        s = s < A_SLOTS ? s : A_SLOTS - 1;                  // slots past the tile re-read its last one
        const int pos = s / PIX;
        ch = s - pos * PIX;
        ch = ch < 4 ? ch : 3;                               // the pad slot re-reads chunk 3 (same cache line)
        row = pos / HW;
        const int rem = pos - row * HW;
        const int par = rem / 17;
        col = 2 * (rem - par * 17) + par;
    };
    for (int c = threadIdx.x; c < COUT; c += 512) lds[BIAS_BASE / 4 + c] = d.bias[c];   // (visible after the prologue barrier)
    {
        // (row, col, chunk) of this lane's slots, packed, parked in LDS: as loop-invariant registers they are 12 VGPRs the stages need
        unsigned *lofs = reinterpret_cast<unsigned *>(const_cast<char *>(ldsb) + LOFS_BASE);
#pragma unroll
        for (int jj = 0; jj < APW; jj++) {
            int row, col, ch;
            slot_of(jj, row, col, ch);
            lofs[jj * 512 + threadIdx.x] = (unsigned)(row | (col << 8) | (ch << 16));
        }
    }
    // The transfer source of (item, slice) = wave-uniform 64-bit base (tile origin, clamped into the plane) + voff[jj] (32-bit, per lane)
    unsigned voff[APW];
    const char *a_base;
    auto tile_offsets = [&](int it) {
        const int pt = it / NOB;
        int ty_, tx_;
        tile_coords(pt, ty_, tx_);
        const int y0 = ty_ * ROWS - d.wino_py + d.off_y, x0 = tx_ * 32 + d.off_x;
        const int yb = clampi(y0, 0, d.in_h - 1), xb = clampi(x0, 0, d.in_w - 1);
        a_base = reinterpret_cast<const char *>(d.in) + ((ABL & 512) ? 0ll : ((long long)yb * d.in_rs + (long long)xb * CIN) * 4);
        const unsigned *lofs = reinterpret_cast<const unsigned *>(ldsb + LOFS_BASE);
        const int rs4 = (int)d.in_rs * 4;   // (a tile spans 10 rows: the byte offsets fit 32 bits for any plane the engine accepts)
#pragma unroll
        for (int jj = 0; jj < APW; jj++) {
            const unsigned pk = lofs[jj * 512 + threadIdx.x];
            const int row = pk & 255, col = (pk >> 8) & 255, ch = pk >> 16;
            const int gy = clampi(y0 + row, 0, d.in_h - 1) - yb;
            const int gx = clampi(x0 + col, 0, d.in_w - 1) - xb;
            voff[jj] = (unsigned)(gy * rs4 + (gx * CIN + 4 * ch) * 4);
        }
    };
    // tile-transfer cursor: the next 16-channel slice to fetch is slice a_lp of item_of(a_n), into tile buffer a_buf (0, 1, 2, 0, ...)
    int a_n = 0, a_lp = 0;
    unsigned a_buf = 0;
    auto dma_a = [&](int jj) {
        const char *sbase = a_base + a_lp * 64;
        const int piece = jj * NW + wave;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(piece < A_PIECES ? lds0 + a_buf * A_BYTES + (unsigned)piece * 1024u : lds0 + DUMP_BASE);
        lds_dma16_s<0>(sbase, voff[jj], dst);
    };
    auto a_advance = [&]() {
        a_buf = a_buf == 2 ? 0u : a_buf + 1u;
        if (++a_lp == NSP) {
            a_lp = 0;
            a_n++;
            tile_offsets(item_of(a_n));
        }
    };
    // U of (32-plane block ob, slice sl_): 16 pieces of 1 KiB, 4 per wave of the group
    const unsigned b_voff = (unsigned)lane * 16u;
    auto dma_b = [&](int ob, int sl_, unsigned slot, int jb) {
        const char *sbase = reinterpret_cast<const char *>(d.wpk) + ((size_t)(ob * NSL + sl_) * 16 + brow * 4) * 1024;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(lds0 + U_BASE + ((unsigned)grp * 2u + slot) * U_BYTES + (unsigned)(brow * 4) * 1024u);
        switch (jb) {
        case 0: lds_dma16_s<0>(sbase, b_voff, dst); break;
        case 1: lds_dma16_s<1024>(sbase, b_voff, dst); break;
        case 2: lds_dma16_s<2048>(sbase, b_voff, dst); break;
        default: lds_dma16_s<3072>(sbase, b_voff, dst); break;
        }
    };

    // ---- fragment addressing ----
    // patch element (r, c) of this lane's block: halo pixel (2w + r, 2t + c) at position (2w + r) * 34 + (c & 1) * 17 + t + (c >> 1) of
    // the tile, channels 2k, 2k+1 of 8-channel sub-slice `sub` = chunk 2 sub + (k >> 1), half k & 1:   base + immediate
    constexpr unsigned P_E = PIX * 16, P_ROW = HW * P_E, P_PAR = 17 * P_E;
    const unsigned pbase = (unsigned)(((2 * brow) * HW + t) * P_E + (k >> 1) * 16 + (k & 1) * 8);
    const unsigned ubase = U_BASE + (unsigned)grp * 2u * U_BYTES + (unsigned)lane * 16u;

    // V of the stage being multiplied / of the next one (roles alternate): [channel 2k + s][xi].  Plain floats and SCALAR additions:
    // v_pk_add_f32 beside MFMAs costs the matrix pipe ~6 cycles apiece (measured: the packed transform was 0.8 of 9.7 ms on 128->128).
    struct VSet { float s[2][16]; };
    VSet va, vb;
    f32x2v raw[16];
    float tq[2][4][4];
    // B^T d B with B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1], columns of d first -- as 64 single additions (op = channel h, output
    // row i, column c), each pinned where it is written: the results are first used a stage later and LLVM otherwise sinks the
    // additions there, out of the MFMA shadows (or re-packs them)
    auto col_op = [&](int h, int i, int c) {
        const float d0 = raw[0 * 4 + c][h], d1 = raw[1 * 4 + c][h], d2 = raw[2 * 4 + c][h], d3 = raw[3 * 4 + c][h];
        float r = i == 0 ? d0 - d2 : i == 1 ? d1 + d2 : i == 2 ? d2 - d1 : d1 - d3;
        asm volatile("" : "+v"(r));
        tq[h][i][c] = r;
    };
    auto row_op = [&](VSet &v, int h, int r, int j) {
        float x = j == 0 ? tq[h][r][0] - tq[h][r][2] : j == 1 ? tq[h][r][1] + tq[h][r][2] : j == 2 ? tq[h][r][2] - tq[h][r][1] : tq[h][r][1] - tq[h][r][3];
        asm volatile("" : "+v"(x));
        v.s[h][r * 4 + j] = x;
    };
    auto transform_cols = [&](int c) {
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int i = 0; i < 4; i++) col_op(h, i, c);
    };
    auto transform_rows = [&](VSet &v, int r) {
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int j = 0; j < 4; j++) row_op(v, h, r, j);
    };

    // ---- prologue: tile slices 0 and 1 of the first item, U(slice 0) of both groups ----
    tile_offsets(item_of(0));
#pragma unroll
    for (int s = 0; s < 2; s++) {
#pragma unroll
        for (int jj = 0; jj < APW; jj++) dma_a(jj);
        a_advance();
    }
#pragma unroll
    for (int jb = 0; jb < 4; jb++) dma_b((item_of(0) % NOB) * 2 + grp, 0, 0, jb);
    W2XC_WAIT_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    {
        // V of the first stage
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < 4; c++)
                raw[r * 4 + c] = *reinterpret_cast<const f32x2v *>(ldsb + pbase + r * P_ROW + (c & 1) * P_PAR + (c >> 1) * P_E);
#pragma unroll
        for (int c = 0; c < 4; c++) transform_cols(c);
#pragma unroll
        for (int r = 0; r < 4; r++) transform_rows(va, r);

        unsigned cbuf = 0;     // tile buffer of the CURRENT stage's 16-channel slice
        unsigned uslot = 0;
        int stamp = 0;
        (void)stamp;
        W16_STAMP(stamp++);
        for (int n = 0; n < nmy; n++) {
            const int item = item_of(n), item_n = item_of(n + 1);
            // The accumulators are DEFINED by the first stage of an item (C = 0) and die in its epilogue
            f32x4 acc[16][2];
            auto stage = [&](auto FIRST, auto ODD, int sl, VSet &vcur, VSet &vnext) {
                constexpr bool first = decltype(FIRST)::value;
                constexpr int odd = decltype(ODD)::value;               // sl & 1 (NSL is even: also the parity of the global stage count)
                constexpr bool issue_a = (odd == 0);                    // even stages carry the tile transfers
                // transfers of this stage: U of the next stage (own group)
                int u_ob = (item % NOB) * 2 + grp, u_sl = sl + 1;
                if (sl == NSL - 1) { u_ob = (item_n % NOB) * 2 + grp; u_sl = 0; }
                // the next stage's patch: even stage -> second half of the current 16-channel slice; odd stage -> first half of the next one
                const unsigned rbuf = odd ? (cbuf == 2 ? 0u : cbuf + 1u) : cbuf;
                const char *pa = ldsb + rbuf * A_BYTES + pbase + (odd ? 0 : 32);
                const char *ua = ldsb + ubase + uslot * U_BYTES;
                f32x4 u[2][2];
                u[0][0] = *reinterpret_cast<const f32x4 *>(ua + 0 * 1024);
                u[0][1] = *reinterpret_cast<const f32x4 *>(ua + 4 * 1024);
                static_for<0, 64>([&](auto SLOT) {
                    constexpr int slot = decltype(SLOT)::value;
                    constexpr int st = slot >> 5, g = slot >> 3, x4 = (slot >> 1) & 3, pt = slot & 1, xi = ((slot >> 3) & 3) * 4 + x4;
                    if constexpr ((ABL & 128) != 0) {
                        if constexpr (first && st == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(acc[xi][pt]) : "v"(u[g & 1][pt][x4]), "v"(vcur.s[st][xi]));
                        else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[xi][pt]) : "v"(u[g & 1][pt][x4]), "v"(vcur.s[st][xi]));
                    } else if constexpr (first && st == 0) {
                        const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
                        acc[xi][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[g & 1][pt][x4], vcur.s[st][xi], z, 0, 0, 0);
                    } else {
                        acc[xi][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[g & 1][pt][x4], vcur.s[st][xi], acc[xi][pt], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // ---- fillers: at most three non-MFMA instructions behind any MFMA ----
                    // U fragments of the next group of 8 MFMAs (group g = (step, xi/4); [step][plane tile][xi/4] KiB in the U slot), read at the
                    // head of group g: 8 MFMAs of cover
                    if constexpr ((slot & 7) < 2 && g < 7 && !(ABL & 32)) {
                        constexpr int gn = g + 1, p = slot & 7;
                        u[gn & 1][p] = *reinterpret_cast<const f32x4 *>(ua + (((gn >> 2) * 2 + p) * 4 + (gn & 3)) * 1024);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // the next stage's patch, column by column, one read per MFMA (slots 2..7, 10..15, 18..21)
                    if constexpr (slot >= 2 && slot < 22 && (slot & 7) >= 2 && !(ABL & 16)) {
                        constexpr int idx = (slot >> 3) * 6 + (slot & 7) - 2;
                        constexpr int r = idx & 3, c = idx >> 2;
                        raw[r * 4 + c] = *reinterpret_cast<const f32x2v *>(pa + r * P_ROW + (c & 1) * P_PAR + (c >> 1) * P_E);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // its transform: 64 scalar additions in FOUR bunches of 16.  Beside the fp32 MFMA a VALU instruction costs 2 cycles
                    // of matrix-pipe time (the two share the fp32 lanes) plus ~4.5 cycles for every MFMA gap that holds any VALU at all
                    // (tools/ubench/mfma_fillers.hip): 4 gaps x 16 beat 32 gaps x 2 by ~6 % of a stage.
                    if constexpr ((ABL & 256) != 0) {   // (the spread schedule, kept for A/B runs)
                        if constexpr (slot >= 22 && slot < 38 && !(ABL & (16 | 64))) {
                            constexpr int o = (slot - 22) * 2;
                            col_op((o >> 2) & 1, o & 3, o >> 3);
                            col_op(((o + 1) >> 2) & 1, (o + 1) & 3, (o + 1) >> 3);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if constexpr (slot >= 38 && slot < 54 && !(ABL & (16 | 64))) {
                            constexpr int o = (slot - 38) * 2;
                            row_op(vnext, (o >> 2) & 1, o >> 3, o & 3);
                            row_op(vnext, ((o + 1) >> 2) & 1, (o + 1) >> 3, (o + 1) & 3);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else if constexpr (!(ABL & (16 | 64))) {
                        if constexpr (slot == 22) { transform_cols(0); transform_cols(1); __builtin_amdgcn_sched_barrier(0); }
                        if constexpr (slot == 30) { transform_cols(2); transform_cols(3); __builtin_amdgcn_sched_barrier(0); }
                        if constexpr (slot == 38) { transform_rows(vnext, 0); transform_rows(vnext, 1); __builtin_amdgcn_sched_barrier(0); }
                        if constexpr (slot == 46) { transform_rows(vnext, 2); transform_rows(vnext, 3); __builtin_amdgcn_sched_barrier(0); }
                    }
                    // transfers: U pieces first, tile pieces last (the stage's closing wait leaves the tile pieces in flight)
                    if constexpr (slot == 3 || slot == 7 || slot == 11 || slot == 15) {
                        if constexpr (!(ABL & 1)) dma_b(u_ob, u_sl, uslot ^ 1u, (slot - 3) >> 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (issue_a && !(ABL & 1024) && (slot == 55 || slot == 57 || slot == 59 || slot == 61)) {
                        if constexpr (!(ABL & 2)) dma_a((slot - 55) >> 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (issue_a && (ABL & 1024) != 0 && (slot == 17 || slot == 19 || slot == 21 || slot == 23)) {
                        if constexpr (!(ABL & 2)) dma_a((slot - 17) >> 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
                if constexpr (issue_a) a_advance();
                W16_STAMP(stamp++);
                // U(next stage) and every older tile piece have landed; this stage's tile pieces (the youngest) may still fly
                if constexpr (issue_a && !(ABL & 2)) W2XC_WAIT_VMCNT(APW);
                else W2XC_WAIT_VMCNT(0);
                W16_STAMP(stamp++);
                if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if constexpr (odd) cbuf = rbuf;
                uslot ^= 1u;
                W16_STAMP(stamp++);
            };
            stage(std::true_type{}, std::integral_constant<int, 0>{}, 0, va, vb);
#pragma unroll 1
            for (int sl = 1; sl < NSL - 1; sl += 2) {
                stage(std::false_type{}, std::integral_constant<int, 1>{}, sl, vb, va);
                stage(std::false_type{}, std::integral_constant<int, 0>{}, sl + 1, va, vb);
            }
            stage(std::false_type{}, std::integral_constant<int, 1>{}, NSL - 1, vb, va);
            {
                // The epilogue at RAISED priority: the two waves of a SIMD leave their last stage together, the older one wins every issue
                // slot, finishes its epilogue first and starts the next item's MFMAs -- under which the younger wave's epilogue then
                // trickles out (s_memtime: 2600 vs 6100 cycles) while the older one ends up waiting at the next barrier.
                if constexpr (!(ABL & 2048)) __builtin_amdgcn_s_setprio(2);
                // (the hazard recogniser does not see inside inline asm: let the last MFMAs drain before VALU reads their results)
                if constexpr ((ABL & 128) != 0) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
                // ---- epilogue: Y = A^T M A (A^T = [1 1 1 0; 0 1 -1 -1]), bias, LeakyReLU, NHWC stores.
                //      C/D of the 16x16 MFMA: lane & 15 = block, register e = plane 4 * (lane >> 4) + e of the plane tile ----
                const int ob = (item % NOB) * 2 + grp, ptile = item / NOB;
                int tile_y, tile_x;
                tile_coords(ptile, tile_y, tile_x);
                const int ty0 = tile_y * ROWS - d.wino_py;
                const int oy = ty0 + 2 * brow, ox = tile_x * 32 + 2 * t;
                float *obase = d.out + (long long)oy * d.out_rs + (long long)ox * COUT + ob * 32 + 4 * k;
                const bool interior = ty0 >= 0 && ty0 + ROWS <= d.out_h && tile_x * 32 + 32 <= d.out_w;   // wave-uniform
                // FUSE: the last layer's weights for this wave's 32 planes as MFMA A operands, w7[pt * 4 + e] = W_last[plane 32 ob + 16 pt +
                // 4 k + e][tap = lane & 15] (0 for taps >= 9): 32 bytes per lane, L2-resident
                f32x4 w7lo = {0.0f, 0.0f, 0.0f, 0.0f}, w7hi = {0.0f, 0.0f, 0.0f, 0.0f};
                f32x4 gacc[2][2];
                if constexpr (FUSE) {
                    const f32x4 *w7 = reinterpret_cast<const f32x4 *>(d.w7pk) + ((size_t)ob * 64 + lane) * 2;
                    w7lo = w7[0];
                    w7hi = w7[1];
#pragma unroll
                    for (int i = 0; i < 2; i++)
#pragma unroll
                        for (int j = 0; j < 2; j++) gacc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                }
#pragma unroll
                for (int pt = 0; pt < 2; pt++) {
                    const f32x4 bq = *reinterpret_cast<const f32x4 *>(ldsb + BIAS_BASE + (ob * 32 + 16 * pt + 4 * k) * 4);
                    f32x4 y[2][2];
                    // two planes at a time: registers 2h, 2h+1 of an accumulator quad are a register PAIR (v_pk_add_f32 without moves)
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        auto m = [&](int xi) { return f32x2v{acc[xi][pt][2 * h], acc[xi][pt][2 * h + 1]}; };
                        const f32x2v b2 = {bq[2 * h], bq[2 * h + 1]};
                        f32x2v tm[2][4];
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            tm[0][j] = m(0 * 4 + j) + m(1 * 4 + j) + m(2 * 4 + j);
                            tm[1][j] = m(1 * 4 + j) - m(2 * 4 + j) - m(3 * 4 + j);
                        }
#pragma unroll
                        for (int i = 0; i < 2; i++) {
                            const f32x2v y0 = tm[i][0] + tm[i][1] + tm[i][2] + b2;
                            const f32x2v y1 = tm[i][1] - tm[i][2] - tm[i][3] + b2;
                            // max(v, 0.1 v) as ONE v_med3_f32(v, 0.1 v, FLT_MAX): fmaxf costs extra canonicalising instructions
                            const f32x2v s0 = y0 * 0.1f, s1 = y1 * 0.1f;
                            y[i][0][2 * h] = __builtin_amdgcn_fmed3f(y0[0], s0[0], 3.402823466e+38f);
                            y[i][0][2 * h + 1] = __builtin_amdgcn_fmed3f(y0[1], s0[1], 3.402823466e+38f);
                            y[i][1][2 * h] = __builtin_amdgcn_fmed3f(y1[0], s1[0], 3.402823466e+38f);
                            y[i][1][2 * h + 1] = __builtin_amdgcn_fmed3f(y1[1], s1[1], 3.402823466e+38f);
                        }
                    }
                    if constexpr (FUSE) {
                        // G[tap][block] += W_last[tap][channel 4 k + e] * act[channel 4 k + e][block], one MFMA per e: the D layout of the
                        // activations (lane quarter k holds planes 4 k .. 4 k + 3) is the B operand of K-step e as it stands
#pragma unroll
                        for (int i = 0; i < 2; i++)
#pragma unroll
                            for (int j = 0; j < 2; j++)
#pragma unroll
                                for (int e = 0; e < 4; e++)
                                    gacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(pt == 0 ? w7lo[e] : w7hi[e], y[i][j][e], gacc[i][j], 0, 0, 0);
                    } else if constexpr ((ABL & 4) != 0) {
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
                if constexpr (FUSE) {
                    // D layout of G: lane & 15 = block, register e = tap 4 k + e (k = 0, 1: four taps; k = 2: tap 8; k = 3: none)
                    float *gbase = d.out + (long long)ob * d.out_ts + (long long)(4 * k) * d.out_gs + (long long)oy * d.out_rs + (long long)ox * d.out_ps;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if (4 * k + e < 9) {
#pragma unroll
                            for (int i = 0; i < 2; i++)
#pragma unroll
                                for (int j = 0; j < 2; j++)
                                    if (interior || (oy + i >= 0 && oy + i < d.out_h && ox + j < d.out_w))
                                        gbase[(long long)e * d.out_gs + (long long)i * d.out_rs + (long long)j * d.out_ps] = gacc[i][j][e];
                        }
                    }
                }
                if constexpr (!(ABL & 2048)) __builtin_amdgcn_s_setprio(0);
                W16_STAMP(stamp++);
            }
        }
    }
    W2XC_WAIT_VMCNT(0);   // drain the speculative transfers before the LDS is released
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
bool w2xc_wino16_supported(int cin, int cout)
{
    return (cin == 32 || cin == 64 || cin == 128) && (cout == 64 || cout == 128);   // (32 output planes: one plane group only -- conv3x3_wino)
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

// The last layer's weights for the FUSE epilogue: dst[32-plane block ob][lane][pt * 4 + e] = w[plane 32 ob + 16 pt + 4 (lane >> 4) + e][tap lane & 15]
// (0 for taps >= 9); w is the one-plane last layer's [1][cin][3][3] (modelHandler.cpp:102).  16 * cin floats.
size_t w2xc_wino16_pack_last_floats(int cin) { return (size_t)16 * cin; }
void w2xc_wino16_pack_last(int cin, const float *w, float *dst)
{
    for (int ob = 0; ob < cin / 32; ob++)
        for (int lane = 0; lane < 64; lane++)
            for (int pt = 0; pt < 2; pt++)
                for (int e = 0; e < 4; e++) {
                    const int plane = 32 * ob + 16 * pt + 4 * (lane >> 4) + e, tap = lane & 15;
                    dst[((size_t)ob * 64 + lane) * 8 + pt * 4 + e] = tap < 9 ? w[(size_t)plane * 9 + tap] : 0.0f;
                }
}

template <int CIN, int COUT, int ABL = 0, bool FUSE = false>
static hipError_t launch_wino16(const W2xcConvDesc &d, hipStream_t stream)
{
    const int tiles_x = (d.out_w + 31) / 32, tiles_y = (d.out_h + (d.wino_py & 1) + 7) / 8;
    const int nitems = tiles_x * tiles_y * (COUT / 64);
    constexpr size_t lds_bytes = 3 * (size_t)(27 * 1024) + 1024 + 4 * (size_t)(16 * 1024) + 4 * 512 * 4 + COUT * 4;   // tile ring + dump + U rings + offset table + bias
    static_assert(lds_bytes <= 160 * 1024, "LDS budget");
    auto kern = conv3x3_wino16<CIN, COUT, ABL, FUSE>;
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
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds_bytes, stream, d, tiles_x, nitems);
    return hipGetLastError();
}

// d.wpk = w2xc_wino16_pack image; NHWC fp32 in / out like W2XC_K_MFMA
hipError_t w2xc_launch_wino16(const W2xcConvDesc &d, hipStream_t stream)
{
    if (d.out_w <= 0 || d.out_h <= 0) return hipSuccess;
    if (d.in_ps != d.cin || d.in_cs != 1 || d.in_shift != 0) return hipErrorInvalidValue;
    if (d.out_terms != 9 && (d.out_ps != d.cout || d.out_cs != 1)) return hipErrorInvalidValue;
    if (d.out_terms == 9) {   // last layer fused: `out` = partial tap planes, d.w7pk = w2xc_wino16_pack_last image
        if (!d.w7pk || (d.in_rs & 3) != 0) return hipErrorInvalidValue;
        switch (d.cin * 1000 + d.cout) {
        case 32064:  return launch_wino16<32, 64, 0, true>(d, stream);
        case 32128:  return launch_wino16<32, 128, 0, true>(d, stream);
        case 64064:  return launch_wino16<64, 64, 0, true>(d, stream);
        case 64128:  return launch_wino16<64, 128, 0, true>(d, stream);
        case 128064: return launch_wino16<128, 64, 0, true>(d, stream);
        case 128128: return launch_wino16<128, 128, 0, true>(d, stream);
        default: return hipErrorInvalidValue;
        }
    }
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
        case 23: return launch_wino16<128, 128, 23>(d, stream);
        case 39: return launch_wino16<128, 128, 39>(d, stream);
        case 55: return launch_wino16<128, 128, 55>(d, stream);
        case 71: return launch_wino16<128, 128, 71>(d, stream);
        case 128: return launch_wino16<128, 128, 128>(d, stream);
        case 135: return launch_wino16<128, 128, 135>(d, stream);
        case 151: return launch_wino16<128, 128, 151>(d, stream);
        case 256: return launch_wino16<128, 128, 256>(d, stream);
        case 2048: return launch_wino16<128, 128, 2048>(d, stream);
        case 512: return launch_wino16<128, 128, 512>(d, stream);
        case 1024: return launch_wino16<128, 128, 1024>(d, stream);
        case 516: return launch_wino16<128, 128, 516>(d, stream);
        case 263: return launch_wino16<128, 128, 263>(d, stream);
        default: break;
    }
#endif
    switch (d.cin * 1000 + d.cout) {
    case 32064:  return launch_wino16<32, 64>(d, stream);
    case 32128:  return launch_wino16<32, 128>(d, stream);
    case 64064:  return launch_wino16<64, 64>(d, stream);
    case 64128:  return launch_wino16<64, 128>(d, stream);
    case 128064: return launch_wino16<128, 64>(d, stream);
    case 128128: return launch_wino16<128, 128>(d, stream);
    default: return hipErrorInvalidValue;
    }
}
