// w2xc_wino4.hip -- conv3x3_wino4: the 3x3 x Cin x Cout contraction of Model::filterWorker
// (/root/reference/src/modelHandler.cpp:117-159) as Winograd F(4x4, 3x3) on v_mfma_f32_16x16x4_f32, two waves per SIMD.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A   per 4x4 output block and 6x6 input patch: 36 positions xi of the transformed domain =
//   36 independent GEMMs  M_xi[o][t] = sum_c U_xi[o][c] V_xi[c][t]  (o = output plane, t = block, c = input plane) -- 36 multiplies
//   for 16 outputs, 2.25 per output against 4 of F(2x2,3x3) (conv3x3_wino16) and 9 of the direct sum.  fp32 throughout; the
//   matrices are the Cook-Toom construction on the interpolation points 0, +-3/4, +-3/2, inf (every entry of B^T and A^T a dyadic rational, exact in fp32):
//     B^T = [81/64 0 -45/16 0 1 0; 0 -27/16 -9/4 3/4 1 0; 0 27/16 -9/4 -3/4 1 0; 0 -27/32 -9/16 3/2 1 0; 0 27/32 -9/16 -3/2 1 0; 0 81/64 0 -45/16 0 1]
//     G   = [64/81 0 0; -128/243 -32/81 -8/27; -128/243 32/81 -8/27; 32/243 16/81 8/27; 32/243 -16/81 8/27; 0 0 1]
//     A^T = [1 1 1 1 1 0; 0 3/4 -3/4 3/2 -3/2 0; 0 9/16 9/16 9/4 9/4 0; 0 27/64 -27/64 27/8 -27/8 1]
//   The points matter (tools/winograd_points.py, numpy): Lavin & Gray's 0, +-1, +-2 (entries up to 5 and 8) put the 7-layer net at 9.5e-6 of the output
//   range against the fp64 truth and use 3.0x the rtol 1e-4 + atol 1e-5 gate on the single-layer standard-normal filter cases (measured on the GPU:
//   2.7e-5 abs at |out| <= 5.7: FAILS); 0, +-1/2, +-3/2: 3.0e-6 / 2.9x; these: 2.6e-6 / 0.8x -- same operation count (symmetric point pairs).
//   (F(2x2): 1.1e-6, the direct fp32 sum 0.9e-6.)
//
//   Work item  16 rows x 32 pixels of output (4 x 8 blocks of 4x4) x 64 output planes.  8 waves: wave (bt, pt) owns block tile bt
//              (16 blocks = block rows 2 bt, 2 bt + 1) x plane tile pt (16 planes) x all 36 xi = 144 accumulators.
//   Stage      one 4-CHANNEL slice = the K of one MFMA: 36 MFMAs per wave (1152 cycles), BOTH operands from LDS in fragment order
//              [xi / 4][tile][lane][xi % 4]: one ds_read_b128 per four xi and operand (A = U, lane = 16 k + o; B = V, lane = 16 k + t).
//   V          is computed ONCE per (block, channel) and shared by the four plane-tile waves through LDS: in stage g the waves with
//              pt == (g + 1 + 2 bt) mod 4 transform the patches of stage g + 1 (one 6x6 patch per lane: 36 ds_read_b32 from the raw tile,
//              144 fma / add, 9 ds_write_b128), interleaved with their own MFMAs.  The item loop exists FOUR times (one copy per
//              transformer phase, unrolled by four stages): see `run` below.
//   LDS        raw[2] x 21 KiB: the 18 x 34 pixel halo tile of an 8-channel slice (two stages), 32 bytes per pixel slot, a row's pixels
//              ordered by column mod 4 (9 slots each) so that the 8 block columns of a patch position are consecutive slots, the two
//              16-byte halves of a slot swapped where bit 2 of its index is set (conflict-free patch reads);
//              U[2] x 36 KiB + V[2] x 18 KiB + 6 KiB of per-lane transfer coordinates + bias = 155 KiB.
//   Transfers  LDS-DMA, SGPR base + 32-bit lane offset: per stage 36 U pieces (one stage ahead), every second stage 21 raw pieces (the
//              slice two slices ahead); U first, raw pieces last: the closing counted vmcnt leaves the raw pieces in flight.
//   Epilogue   Y = A^T M A per output-row pair, bias, LeakyReLU, 16-byte NHWC stores.
//   Banding    blocks sit on rows = 0 mod 4 of the layer's whole output (W2xcConvDesc::wino_py = first row mod 4): interior blocks do not
//              depend on the banding, blocks that straddle a band edge do at rounding level (every output of a block sees all 36 patch
//              values; F(2x2) outputs do not) -- one of the two reasons this kernel is opt-in.
// Measured (round 3, 2160x3840): 128->128 7.1 ms (conv3x3_wino16 8.8), frame 16.4 ms against 19.6; 0.52-0.55 of the fp32 MFMA rate on the multiplies
// it issues -- what the rest is: DESIGN.md 3, profiles/r3_sweeps.log block 20.
#include "w2xc_kernels.h"
#include "w2xc_device.h"

#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <type_traits>

#ifdef W4_TIMING
// tools/ubench/wino4_timing.hip: s_memtime stamps of waves 0 and 4 of workgroup 0 -- per stage (before the closing wait, after it, after the
// barrier) and after every epilogue -- [wave >> 2][index]
__device__ unsigned long long w4_stamps[2][8192];
#define W4_STAMP(idx) do { const int i_ = (idx); if (blockIdx.x == 0 && pt == 0 && lane == 0 && i_ < 8192) w4_stamps[bt][i_] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W4_STAMP(idx) do { } while (0)
#endif
#ifndef W4_ABL
#define W4_ABL 0   // timing-only ablations (wrong results): 1 no transform arithmetic | 2 no patch reads | 4 no V writes | 16 no U transfers | 32 no raw transfers | 64 no epilogue stores
#endif

namespace {

// y = B^T x for a 6-vector, in place (14 fma / mul / add)
static __device__ __forceinline__ void bt6(float &x0, float &x1, float &x2, float &x3, float &x4, float &x5)
{
    const float y0 = __builtin_fmaf(-2.8125f, x2, __builtin_fmaf(1.265625f, x0, x4));
    const float p = __builtin_fmaf(-2.25f, x2, x4), q = __builtin_fmaf(-1.6875f, x1, 0.75f * x3);
    const float u = __builtin_fmaf(-0.5625f, x2, x4), v = __builtin_fmaf(-0.84375f, x1, 1.5f * x3);
    const float y5 = __builtin_fmaf(-2.8125f, x3, __builtin_fmaf(1.265625f, x1, x5));
    x0 = y0;
    x1 = p + q;
    x2 = p - q;
    x3 = u + v;
    x4 = u - v;
    x5 = y5;
}

// y = A^T m for a 6-vector (12 fma / mul / add)
static __device__ __forceinline__ void at6(float m0, float m1, float m2, float m3, float m4, float m5, float &y0, float &y1, float &y2, float &y3)
{
    const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
    y0 = m0 + s1 + s2;
    y1 = __builtin_fmaf(1.5f, d2, 0.75f * d1);
    y2 = __builtin_fmaf(2.25f, s2, 0.5625f * s1);
    y3 = __builtin_fmaf(3.375f, d2, __builtin_fmaf(0.421875f, d1, m5));
}

}   // namespace

template <int CIN, int COUT>
__global__ void __launch_bounds__(512, 2) conv3x3_wino4(W2xcConvDesc d, int tiles_x, int nitems)
{
    constexpr int ROWS = 16, HW = 34, HH = ROWS + 2;
    constexpr int NST = CIN / 4;                            // stages (4-channel slices) per item
    constexpr int NSP = CIN / 8;                            // 8-channel raw slices per item
    constexpr int NOB = COUT / 64;                          // 64-plane blocks
    constexpr int NW = 8;
    constexpr int RSLOT = 36;                               // pixel slots per tile row: 4 column residues x 9
    constexpr int RAW_SLOTS = HH * RSLOT;                   // 648
    constexpr int RAW_PIECES = (RAW_SLOTS * 2 + 63) / 64;   // 21 pieces of 1 KiB (64 lanes x 16 bytes = 32 pixel slots)
    constexpr int RPW = 3;                                  // pieces per wave (pieces >= 21 repeat the last one)
    constexpr unsigned RAW_BYTES = RAW_PIECES * 1024;
    constexpr unsigned U_BASE = 2 * RAW_BYTES, U_BYTES = 36 * 1024;
    constexpr unsigned V_BASE = U_BASE + 2 * U_BYTES, V_BYTES = 18 * 1024;
    constexpr unsigned LOFS_BASE = V_BASE + 2 * V_BYTES;
    constexpr unsigned BIAS_BASE = LOFS_BASE + RPW * 512 * 4;
    static_assert(CIN % 16 == 0 && COUT % 64 == 0 && NST % 4 == 0, "planes");
    constexpr int STRIP = 16;
    const int tiles_y = nitems / (NOB * tiles_x);
    auto tile_coords = [&](int pt_, int &ty_, int &tx_) {     // strips of 16 tiles, row by row inside a strip (see conv3x3_wino16)
        const int per_strip = STRIP * tiles_y;
        int sidx = pt_ / per_strip;
        const int nfull = tiles_x / STRIP;
        if (sidx > nfull) sidx = nfull;
        const int wid = sidx < nfull ? STRIP : tiles_x - nfull * STRIP;
        const int q = pt_ - sidx * per_strip;
        ty_ = q / wid;
        tx_ = sidx * STRIP + (q - ty_ * wid);
    };
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    char *ldsb = reinterpret_cast<char *>(lds);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pt = wave & 3, bt = wave >> 2;
    const int t = lane & 15, k = lane >> 4;

    const int xcd = blockIdx.x & 7, per = gridDim.x >> 3;
    const int cq = nitems >> 3, cr = nitems & 7;
    const int chunk_begin = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
    const int chunk_end = chunk_begin + cq + (xcd < cr ? 1 : 0);
    const int item0 = chunk_begin + (blockIdx.x >> 3);
    if (item0 >= chunk_end) return;
    const int nmy = (chunk_end - item0 + per - 1) / per;
    auto item_of = [&](int n) { return item0 + (n < nmy ? n : nmy - 1) * per; };

    for (int c = threadIdx.x; c < COUT; c += 512) lds[BIAS_BASE / 4 + c] = d.bias[c];   // (visible after the prologue barrier)

    // ---- raw tile transfers: element e = piece * 64 + lane -> pixel slot e >> 1, 16-byte half e & 1 of its 8 channels ----
    {
        // (row, col, half) of this lane's elements, packed, parked in LDS: as loop-invariant registers they are 9 VGPRs the stages need
        unsigned *lofs = reinterpret_cast<unsigned *>(ldsb + LOFS_BASE);
#pragma unroll
        for (int jj = 0; jj < RPW; jj++) {
            int piece = jj * NW + wave;
            piece = piece < RAW_PIECES ? piece : RAW_PIECES - 1;
            const int e = piece * 64 + lane;
            int slot = e >> 1;
            slot = slot < RAW_SLOTS ? slot : RAW_SLOTS - 1;
            const int row = slot / RSLOT, rem = slot - row * RSLOT;
            const int res = rem / 9, idx = rem - res * 9;
            int col = 4 * idx + res;
            col = col < HW ? col : HW - 1;
            // the two 16-byte halves of a pixel slot are SWAPPED where bit 2 of the slot's index in its residue group is set: the 8 block columns x 4
            // channels of a patch position then hit 32 different banks (bank = 8 slot + 4 half + k)
            lofs[jj * 512 + threadIdx.x] = (unsigned)(row | (col << 8) | ((((e & 1) ^ (idx >> 2)) & 1) << 16));
        }
    }
    unsigned voff[RPW];
    const char *a_base;
    auto tile_offsets = [&](int it) {
        int ty_, tx_;
        tile_coords(it / NOB, ty_, tx_);
        const int y0 = ty_ * ROWS - d.wino_py + d.off_y, x0 = tx_ * 32 + d.off_x;
        const int yb = clampi(y0, 0, d.in_h - 1), xb = clampi(x0, 0, d.in_w - 1);
        a_base = reinterpret_cast<const char *>(d.in) + ((long long)yb * d.in_rs + (long long)xb * CIN) * 4;
        const int rs4 = (int)d.in_rs * 4;
        const unsigned *lofs = reinterpret_cast<const unsigned *>(ldsb + LOFS_BASE);
#pragma unroll
        for (int jj = 0; jj < RPW; jj++) {
            const unsigned pk = lofs[jj * 512 + threadIdx.x];
            const int row = pk & 255, col = (pk >> 8) & 255, half = pk >> 16;
            const int gy = clampi(y0 + row, 0, d.in_h - 1) - yb;
            const int gx = clampi(x0 + col, 0, d.in_w - 1) - xb;
            voff[jj] = (unsigned)(gy * rs4 + (gx * CIN + 4 * half) * 4);
        }
    };
    // raw cursor: the next 8-channel slice to fetch is slice r_lp of item_of(r_n), into raw buffer r_buf
    int r_n = 0, r_lp = 0;
    unsigned r_buf = 0;
    auto dma_raw = [&](int jj) {
        const char *sbase = a_base + r_lp * 32;
        int piece = jj * NW + wave;
        piece = piece < RAW_PIECES ? piece : RAW_PIECES - 1;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(lds0 + r_buf * RAW_BYTES + (unsigned)piece * 1024u);
        lds_dma16_s<0>(sbase, voff[jj], dst);
    };
    auto raw_advance = [&]() {
        r_buf ^= 1u;
        if (++r_lp == NSP) {
            r_lp = 0;
            r_n++;
            tile_offsets(item_of(r_n));
        }
    };
    // U of (64-plane block ob, stage s_): 36 pieces of 1 KiB (one per xi); wave w sends xi = w, w + 8, w + 16, w + 24 and (w < 4) 32 + w
    const unsigned b_voff = (unsigned)lane * 16u;
    auto dma_u = [&](int ob, int s_, unsigned slot, int xi) {
        const char *sbase = reinterpret_cast<const char *>(d.wpk) + ((size_t)(ob * NST + s_) * 36 + xi) * 1024;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(lds0 + U_BASE + slot * U_BYTES + (unsigned)xi * 1024u);
        lds_dma16_s<0>(sbase, b_voff, dst);
    };

    // ---- addressing ----
    // MFMA operands: lane-linear dwords
    const unsigned ua0 = U_BASE + (unsigned)pt * 1024u + (unsigned)lane * 16u;    // + slot * U_BYTES + (xi / 4) * 4096: four xi per b128
    const unsigned va0 = V_BASE + (unsigned)bt * 1024u + (unsigned)lane * 16u;    // + slot * V_BYTES + (xi / 4) * 2048
    // transformer lane (r, kk, c) = block (block row 2 bt + r, column c), channel kk: patch pixel (i, j) sits in raw slot
    // (4 (2 bt + r) + i) * 36 + (j & 3) * 9 + c + (j >> 2)
    const int tr_r = lane >> 5, tr_k = (lane >> 3) & 3, tr_c = lane & 7;
    const unsigned tr_rd = (unsigned)(((4 * (2 * bt + tr_r)) * RSLOT + tr_c) * 32 + tr_k * 4);          // + buffer + immediates + the swizzled half:
    const unsigned tr_sw0 = (unsigned)(((tr_c >> 2) & 1) * 16), tr_sw1 = (unsigned)((((tr_c + 1) >> 2) & 1) * 16);   // patch columns 0..3 / 4, 5
    const unsigned tr_wr = V_BASE + (unsigned)bt * 1024u + (unsigned)((tr_k * 16 + tr_r * 8 + tr_c) * 16);  // + slot * V_BYTES + (xi / 4) * 2048

    // the transform of one patch: reads (slots 0..11 of a stage), columns (12..17), rows + writes (18..23)
    float dd[36];
    // (indices arrive as integral constants: register arrays indexed through a run-time lambda parameter end up in scratch)
    auto tr_read = [&](const char *src, const char *src1, auto Q) {   // three patch elements per call, Q = 0..11; src / src1: columns 0..3 / 4, 5
        static_for<0, 3>([&](auto E3) {
            constexpr int e = decltype(Q)::value * 3 + decltype(E3)::value, i = e / 6, j = e % 6;
            if constexpr ((W4_ABL & 2) != 0) dd[e] = (float)e;
            else {
                // inline asm: as C++ loads these reads are merged into ds_read2_b32 pairs whose 8-bit offsets need extra address registers, which the
                // compiler hoists and then SPILLS -- and a scratch reload waits with vmcnt(0), i.e. for every transfer in flight (2000 cycles per
                // transforming stage, measured).  The compiler does not count these reads in lgkmcnt: tr_wait() below is their wait.
                constexpr int off = i * (RSLOT * 32) + ((j & 3) * 9 + (j >> 2)) * 32;
                const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)(j < 4 ? src : src1);
                float v;
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off));
                dd[e] = v;
            }
        });
    };
    auto tr_wait = [&]() {   // every patch read has returned (and the compiler sees the values as defined HERE)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(dd[0]), "+v"(dd[1]), "+v"(dd[2]), "+v"(dd[3]), "+v"(dd[4]), "+v"(dd[5]), "+v"(dd[6]), "+v"(dd[7]), "+v"(dd[8]), "+v"(dd[9]), "+v"(dd[10]),
                       "+v"(dd[11]), "+v"(dd[12]), "+v"(dd[13]), "+v"(dd[14]), "+v"(dd[15]), "+v"(dd[16]), "+v"(dd[17]));
        asm volatile(""
                     : "+v"(dd[18]), "+v"(dd[19]), "+v"(dd[20]), "+v"(dd[21]), "+v"(dd[22]), "+v"(dd[23]), "+v"(dd[24]), "+v"(dd[25]), "+v"(dd[26]), "+v"(dd[27]),
                       "+v"(dd[28]), "+v"(dd[29]), "+v"(dd[30]), "+v"(dd[31]), "+v"(dd[32]), "+v"(dd[33]), "+v"(dd[34]), "+v"(dd[35]));
    };
    auto tr_col = [&](auto J) {
        constexpr int j = decltype(J)::value;
        if constexpr (!(W4_ABL & 1)) bt6(dd[0 * 6 + j], dd[1 * 6 + j], dd[2 * 6 + j], dd[3 * 6 + j], dd[4 * 6 + j], dd[5 * 6 + j]);
    };
    auto tr_row = [&](char *dst, auto I) {
        constexpr int i = decltype(I)::value;
        if constexpr (!(W4_ABL & 1)) bt6(dd[i * 6 + 0], dd[i * 6 + 1], dd[i * 6 + 2], dd[i * 6 + 3], dd[i * 6 + 4], dd[i * 6 + 5]);
        // the quads of four consecutive xi that this row completes: 4 q + 3 <= 6 i + 5 and not already complete after row i - 1
        static_for<(i == 0 ? 0 : (6 * i - 4) / 4 + 1), (6 * i + 2) / 4 + 1>([&](auto Q4) {
            constexpr int q4 = decltype(Q4)::value;
            if constexpr (!(W4_ABL & 4))
                *reinterpret_cast<f32x4 *>(dst + q4 * 2048) = f32x4{dd[4 * q4], dd[4 * q4 + 1], dd[4 * q4 + 2], dd[4 * q4 + 3]};
        });
    };

    // The wave that transforms the patches of stage g + 1 (in stage g) is the one with pt == (g + 1 + 2 bt) mod 4, i.e. every wave in every
    // fourth stage, at its own phase PH = (pt - 1 - 2 bt) mod 4.  A run-time "is it my turn" around the two stage bodies joins 144 accumulators
    // in phi nodes after every stage and the register allocator gives up (150 spilled registers); so the whole item loop exists four times,
    // unrolled by four stages with the transforming stage fixed at compile time, and a wave picks its copy once.
    auto run = [&](auto PH_) {
    constexpr int PH = decltype(PH_)::value;
    // ---- prologue: raw slices 0 and 1 of the first item, U(stage 0), then V(stage 0) ----
    tile_offsets(item_of(0));
#pragma unroll
    for (int s = 0; s < 2; s++) {
#pragma unroll
        for (int jj = 0; jj < RPW; jj++) dma_raw(jj);
        raw_advance();
    }
    {
        const int ob0 = item_of(0) % NOB;
#pragma unroll
        for (int q = 0; q < 4; q++) dma_u(ob0, 0, 0, q * 8 + wave);
        if (wave < 4) dma_u(ob0, 0, 0, 32 + wave);
    }
    W2XC_WAIT_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (PH == 3) {   // V of the first stage (stage -1 = 3 mod 4)
        const char *src = ldsb + tr_rd + tr_sw0, *src1 = ldsb + tr_rd + tr_sw1;
        char *dst = ldsb + tr_wr;
        static_for<0, 12>([&](auto Q) { tr_read(src, src1, Q); });
        tr_wait();
        static_for<0, 6>([&](auto J) { tr_col(J); });
        static_for<0, 6>([&](auto I) { tr_row(dst, I); });
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    unsigned par = 0;         // parity of the global stage count: U / V buffer of the CURRENT stage
    int stamp = 0;
    (void)stamp;
    W4_STAMP(stamp++);
    unsigned r_buf_cur = 0;   // raw buffer of the current stage's 8-channel slice
    for (int n = 0; n < nmy; n++) {
        const int item = item_of(n), item_n = item_of(n + 1);
        f32x4 acc[36];
#pragma unroll
        for (int xi = 0; xi < 36; xi++) acc[xi] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

        // one stage; ODD = stage parity (odd stages carry the raw transfers), TR = this wave transforms the next stage's patches
        auto stage = [&](auto ODD_, auto TR_, int s) {
            constexpr bool odd = decltype(ODD_)::value;
            constexpr bool tr = decltype(TR_)::value;
            int u_ob = item % NOB, u_s = s + 1;
            if (s == NST - 1) { u_ob = item_n % NOB; u_s = 0; }
            const unsigned nxt = par ^ 1u;
            const char *ua = ldsb + ua0 + par * U_BYTES;
            const char *va = ldsb + va0 + par * V_BYTES;
            // next stage's patches: raw slice of global stage g + 1; stage parity odd -> g + 1 even -> first half of the NEXT raw buffer
            const unsigned rb = odd ? (r_buf_cur ^ 1u) : r_buf_cur;
            const char *src = ldsb + rb * RAW_BYTES + tr_rd + ((odd ? 0u : 16u) ^ tr_sw0);
            const char *src1 = ldsb + rb * RAW_BYTES + tr_rd + ((odd ? 0u : 16u) ^ tr_sw1);
            char *dst = ldsb + tr_wr + nxt * V_BYTES;
            // operands of four xi per ds_read_b128; the quads of the next four xi are read while these four multiply
            f32x4 a4[2], b4[2];
            a4[0] = *reinterpret_cast<const f32x4 *>(ua);
            b4[0] = *reinterpret_cast<const f32x4 *>(va);
            static_for<0, 36>([&](auto XI) {
                constexpr int xi = decltype(XI)::value;
                if constexpr ((xi & 3) == 0 && xi + 4 < 36) {
                    a4[((xi >> 2) + 1) & 1] = *reinterpret_cast<const f32x4 *>(ua + ((xi >> 2) + 1) * 4096);
                    b4[((xi >> 2) + 1) & 1] = *reinterpret_cast<const f32x4 *>(va + ((xi >> 2) + 1) * 2048);
                    __builtin_amdgcn_sched_barrier(0);
                }
                acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[(xi >> 2) & 1][xi & 3], b4[(xi >> 2) & 1][xi & 3], acc[xi], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // transfers: U pieces first, raw pieces last
                if constexpr ((xi == 1 || xi == 3 || xi == 5 || xi == 7) && !(W4_ABL & 16)) {
                    dma_u(u_ob, u_s, nxt, ((xi - 1) >> 1) * 8 + wave);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (xi == 9 && !(W4_ABL & 16)) {
                    if (wave < 4) dma_u(u_ob, u_s, nxt, 32 + wave);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (odd && (xi == 29 || xi == 31 || xi == 33) && !(W4_ABL & 32)) {
                    dma_raw((xi - 29) >> 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (tr) {
                    // the whole transform in the first half of the stage, while the partner wave of the SIMD still issues MFMAs: s_memtime per slot
                    // showed the row pass at slots 18..23 -- after the partner had finished -- costing 1950 cycles, the identical column pass at 12..17 360
                    if constexpr (xi < 6) {
                        tr_read(src, src1, std::integral_constant<int, 2 * xi>{});
                        tr_read(src, src1, std::integral_constant<int, 2 * xi + 1>{});
                        __builtin_amdgcn_sched_barrier(0);
                    } else if constexpr (xi < 12) {
                        if constexpr (xi == 6) tr_wait();
                        tr_col(std::integral_constant<int, xi - 6>{});
                        __builtin_amdgcn_sched_barrier(0);
                    } else if constexpr (xi < 18) {
                        tr_row(dst, std::integral_constant<int, xi - 12>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            });
            if constexpr (odd) {
                raw_advance();
                r_buf_cur ^= 1u;
            }
            W4_STAMP(stamp++);
            if constexpr (odd && !(W4_ABL & 32)) W2XC_WAIT_VMCNT(RPW);     // U(next stage) has landed; this stage's raw pieces (the youngest) may still fly
            else W2XC_WAIT_VMCNT(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            W4_STAMP(stamp++);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            par = nxt;
            W4_STAMP(stamp++);
        };
#pragma unroll 1
        for (int s = 0; s < NST; s += 4) {   // (NST is a multiple of 4: the global stage count mod 4 = s mod 4)
            stage(std::false_type{}, std::integral_constant<bool, PH == 0>{}, s);
            stage(std::true_type{}, std::integral_constant<bool, PH == 1>{}, s + 1);
            stage(std::false_type{}, std::integral_constant<bool, PH == 2>{}, s + 2);
            stage(std::true_type{}, std::integral_constant<bool, PH == 3>{}, s + 3);
        }
        {
            // ---- epilogue: Y = A^T M A, bias, LeakyReLU, NHWC stores.  C/D of the 16x16 MFMA: lane & 15 = block, register e = plane
            //      4 (lane >> 4) + e of the plane tile ----
            __builtin_amdgcn_s_setprio(2);
            const int ob = item % NOB;
            int tile_y, tile_x;
            tile_coords(item / NOB, tile_y, tile_x);
            const int ty0 = tile_y * ROWS - d.wino_py;
            const int oy = ty0 + 4 * (2 * bt + (t >> 3)), ox = tile_x * 32 + 4 * (t & 7);
            float *obase = d.out + (long long)oy * d.out_rs + (long long)ox * COUT + ob * 64 + pt * 16 + 4 * k;
            const bool interior = ty0 >= 0 && ty0 + ROWS <= d.out_h && tile_x * 32 + 32 <= d.out_w;   // wave-uniform
            const f32x4 bq = *reinterpret_cast<const f32x4 *>(ldsb + BIAS_BASE + (ob * 64 + pt * 16 + 4 * k) * 4);
            // Two output ROWS of the block at a time, all four planes of a lane: 16-byte stores (a lane's four planes of a pixel; the four lanes
            // of a block cover 64 contiguous bytes) -- 16 store instructions per lane instead of the 32 eight-byte ones of the first version,
            // whose 8 x 32 scattered stores per item kept the CU's address unit busy for 16000 cycles.  The row transform of a column is done
            // per row PAIR (7 operations instead of 10 for all four rows): 48 + 32 live values beside the 144 accumulators.
#pragma unroll
            for (int rp = 0; rp < 2; rp++) {
                float tm[2][6][4];   // [row of the pair][column j][plane e]
#pragma unroll
                for (int e = 0; e < 4; e++)
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        const float m0 = acc[0 * 6 + j][e], m1 = acc[1 * 6 + j][e], m2 = acc[2 * 6 + j][e], m3 = acc[3 * 6 + j][e], m4 = acc[4 * 6 + j][e],
                                    m5 = acc[5 * 6 + j][e];
                        const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
                        if (rp == 0) {
                            tm[0][j][e] = m0 + s1 + s2;
                            tm[1][j][e] = __builtin_fmaf(1.5f, d2, 0.75f * d1);
                        } else {
                            tm[0][j][e] = __builtin_fmaf(2.25f, s2, 0.5625f * s1);
                            tm[1][j][e] = __builtin_fmaf(3.375f, d2, __builtin_fmaf(0.421875f, d1, m5));
                        }
                    }
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    const int i = 2 * rp + rr;
                    f32x4 y[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        float y0, y1, y2, y3;
                        at6(tm[rr][0][e], tm[rr][1][e], tm[rr][2][e], tm[rr][3][e], tm[rr][4][e], tm[rr][5][e], y0, y1, y2, y3);
                        const float v0 = y0 + bq[e], v1 = y1 + bq[e], v2 = y2 + bq[e], v3 = y3 + bq[e];
                        y[0][e] = __builtin_amdgcn_fmed3f(v0, 0.1f * v0, 3.402823466e+38f);
                        y[1][e] = __builtin_amdgcn_fmed3f(v1, 0.1f * v1, 3.402823466e+38f);
                        y[2][e] = __builtin_amdgcn_fmed3f(v2, 0.1f * v2, 3.402823466e+38f);
                        y[3][e] = __builtin_amdgcn_fmed3f(v3, 0.1f * v3, 3.402823466e+38f);
                    }
                    if constexpr ((W4_ABL & 64) != 0) {
                        if (y[0][0] == 12345.678f) *reinterpret_cast<f32x4 *>(obase) = y[0] + y[1] + y[2] + y[3];
                    } else if (interior) {
#pragma unroll
                        for (int j = 0; j < 4; j++) *reinterpret_cast<f32x4 *>(obase + (long long)i * d.out_rs + j * COUT) = y[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if (oy + i >= 0 && oy + i < d.out_h && ox + j < d.out_w) *reinterpret_cast<f32x4 *>(obase + (long long)i * d.out_rs + j * COUT) = y[j];
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
            W4_STAMP(stamp++);
        }
    }
    W2XC_WAIT_VMCNT(0);   // drain the speculative transfers before the LDS is released
    };
    switch ((pt - 1 - 2 * bt) & 3) {
    case 0: run(std::integral_constant<int, 0>{}); break;
    case 1: run(std::integral_constant<int, 1>{}); break;
    case 2: run(std::integral_constant<int, 2>{}); break;
    default: run(std::integral_constant<int, 3>{}); break;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
bool w2xc_wino4_supported(int cin, int cout)
{
    return (cin == 32 || cin == 64 || cin == 128) && (cout == 64 || cout == 128);
}

template <int CIN, int COUT>
static hipError_t launch_wino4(const W2xcConvDesc &d, hipStream_t stream)
{
    const int tiles_x = (d.out_w + 31) / 32, tiles_y = (d.out_h + (d.wino_py & 3) + 15) / 16;
    const int nitems = tiles_x * tiles_y * (COUT / 64);
    constexpr size_t lds_bytes = 2 * (size_t)(21 * 1024) + 2 * (size_t)(36 * 1024) + 2 * (size_t)(18 * 1024) + 3 * 512 * 4 + COUT * 4;   // raw + U + V + offset table + bias
    static_assert(lds_bytes <= 160 * 1024, "LDS budget");
    auto kern = conv3x3_wino4<CIN, COUT>;
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

// d.wpk = w2xc_wino4_pack image; NHWC fp32 in / out like W2XC_K_MFMA; d.wino_py = first output row mod 4
hipError_t w2xc_launch_wino4(const W2xcConvDesc &d, hipStream_t stream)
{
    if (d.out_w <= 0 || d.out_h <= 0) return hipSuccess;
    if (d.in_ps != d.cin || d.in_cs != 1 || d.in_shift != 0) return hipErrorInvalidValue;
    if (d.out_ps != d.cout || d.out_cs != 1) return hipErrorInvalidValue;
    if ((d.in_rs & 3) != 0 || (d.out_rs & 3) != 0) return hipErrorInvalidValue;   // 16-byte accesses
    switch (d.cin * 1000 + d.cout) {
    case 32064:  return launch_wino4<32, 64>(d, stream);
    case 32128:  return launch_wino4<32, 128>(d, stream);
    case 64064:  return launch_wino4<64, 64>(d, stream);
    case 64128:  return launch_wino4<64, 128>(d, stream);
    case 128064: return launch_wino4<128, 64>(d, stream);
    case 128128: return launch_wino4<128, 128>(d, stream);
    default: return hipErrorInvalidValue;
    }
}
