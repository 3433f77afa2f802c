// w2xc_wino4.hip -- conv3x3_wino4: the 3x3 x Cin x Cout contraction of Model::filterWorker
// (/root/reference/src/modelHandler.cpp:117-159) as Winograd F(4x4, 3x3) on v_mfma_f32_16x16x4_f32, two waves per SIMD,
// on PLANAR activations (one H x W fp32 plane per channel -- the reference's own std::vector<cv::Mat> layout).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A   per 4x4 output block and 6x6 input patch: 36 positions xi of the transformed domain =
//   36 independent GEMMs  M_xi[o][t] = sum_c U_xi[o][c] V_xi[c][t]  (o = output plane, t = block, c = input plane): 2.25 multiplies
//   per output.  fp32 throughout; Cook-Toom on the points 0, +-3/4, +-3/2, inf (tools/winograd_points.py: every entry a dyadic rational):
//     B^T = [81/64 0 -45/16 0 1 0; 0 -27/16 -9/4 3/4 1 0; 0 27/16 -9/4 -3/4 1 0; 0 -27/32 -9/16 3/2 1 0; 0 27/32 -9/16 -3/2 1 0; 0 81/64 0 -45/16 0 1]
//     G   = [64/81 0 0; -128/243 -32/81 -8/27; -128/243 32/81 -8/27; 32/243 16/81 8/27; 32/243 -16/81 8/27; 0 0 1]
//     A^T = [1 1 1 1 1 0; 0 3/4 -3/4 3/2 -3/2 0; 0 9/16 9/16 9/4 9/4 0; 0 27/64 -27/64 27/8 -27/8 1]
//
//   Work item  16 rows x 32 pixels of output (4 x 8 blocks of 4x4) x 64 output planes.  8 waves: wave (bt, pt) owns block tile bt
//              (block rows 2 bt, 2 bt + 1) x plane tile pt (16 planes) x all 36 xi = 144 accumulators.
//   Stage      one 4-CHANNEL slice = the K of one MFMA: 36 MFMAs per wave, both operands from LDS in fragment order
//              [xi / 4][tile][lane][xi % 4] (one ds_read_b128 per four xi and operand, read two groups of four MFMAs ahead; A = U, lane = 16 k + o;
//              B = V, lane = 16 k + t).  The stage's closing wait + barrier sit in FRONT of its last four MFMAs: behind the barrier a wave reads the
//              first operands of the next stage and still has four MFMAs to issue while they arrive.  An item's first stage is its own copy of the
//              stage body: its MFMAs take 0 as accumulator input (no zeroing of 144 registers per wave and item).
//   V          is computed once per (block, channel) and shared by the four plane-tile waves through LDS.  ONE wave transforms a 4-channel slice of
//              its block tile in FOUR QUARTERS over four consecutive stages, the 36 values in REGISTERS in between (Q0 raw patch + row pass of rows
//              0..2 | Q1 rows 3..5 | Q2 column pass of columns 0..2 | Q3 columns 3..5 + nine ds_write_b128 of V): every wave carries the same 42 VALU
//              instructions in every stage, the four waves of a block tile are one quarter apart, the two waves of a SIMD two.  The pipeline runs
//              across items; the waves in mid-transform at an item's end park their 36 values in LDS across the epilogue.  V is two slots.
//              The MFMAs are asm with the accumulator TIED: as builtins the allocator gave three of four a destination other than their
//              accumulator input, the accumulators migrated through the file and some were spilled inside the stages.
//   LDS        raw[3] x 11 KiB: the 18 x 36 pixel halo tile of a 4-channel slice as 16-byte chunks (channel kk, row R, pixel quad q) at
//              chunk index kk * 168 + R * 9 + q (the stride 168 = 8 mod 16 makes the b128 patch reads conflict-free);
//              U[2] x 36 KiB + V[2] x 18 KiB + 18 KiB (with U slot 1 the parking area of an epilogue) + bias = 159.5 KiB.
//   Transfers  LDS-DMA (global_load_lds_dwordx4), SGPR base + 32-bit lane offset: per stage 36 U pieces (one stage ahead) and 11 raw
//              pieces (the slice Q0 reads two stages later; the closing wait of a stage leaves its own raw pieces in flight), all issued by the
//              four OLDER waves: they win the matrix pipe's arbitration and have the time (W4_DMA4).  A lane's 16 bytes are four
//              consecutive pixels of one plane row: whole 128-byte lines (the engine gives planar rows a stride of roundup32(w) floats).
//   32 planes in  (IN_NHWC) the producers conv3x3_first / conv3x3_wino write NHWC pixels of one 128-byte line: a raw chunk is then the four
//              channels of ONE pixel, the pixels of a row grouped by column mod 4 (conflict-free ds_read_b32 of a lane's channel).
//   Epilogue   Y = A^T M A per output-row pair, bias, LeakyReLU; planar out: one 16-byte store = four pixels of a plane row, 8 lanes = one
//              128-byte line (NHWC out for a consumer that wants it).  FUSE7: the model's one-plane LAST layer on the activations just
//              computed ("taps as rows" on the MFMA), the four plane tiles of a block summed through the idle V slot: 9 partial tap planes
//              per 64-plane block leave the chip (72 B per pixel instead of 512), conv3x3_last_gather finishes.
//   Edges      rows are clamped (replicate) in the transfer addresses; patch columns >= in_w are ZEROED in the row pass (they only reach
//              outputs >= out_w, and what is in memory there is not defined): results do not depend on memory contents outside the plane.
//   Banding    blocks sit on rows = 0 mod 4 of the layer's whole output (W2xcConvDesc::wino_py = first row mod 4); run_rows' four-rows-per-layer
//              band geometry makes every region edge that is not a plane edge a block edge: bit-identical results across bandings.
// Measured (round 4, 2160x3840, one MI355X): 128 -> 128 6.4-6.8 ms (round 3's NHWC kernel: 7.5 on the same box), frame 14.4-14.9 ms (16.4).  s_memtime: a
// stage takes ~3350 cycles (2304 = the matrix pipe's time for the 72 MFMAs of a SIMD; a synthetic loop of the same shape without transfers: 2670), an
// item's boundary another ~8k of its 116k (profiles/r4_sweeps.log).
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
#ifndef W4P_T0
#define W4P_T0 3     // MFMA slot of a stage behind which a wave's transform arithmetic starts (its LDS reads sit behind slot 0)
#endif
#ifndef W4_PF
#define W4_PF 2      // operand look-ahead in groups of four MFMAs (1: two register buffers, 2: three)
#endif
#ifndef W4_DMA4
#define W4_DMA4 1    // 1: the four OLDER waves (block tile 0) issue every transfer (nine U pieces + two / three raw pieces each per stage), the younger four none
#endif
#ifndef W4_ABL
#define W4_ABL 0   // timing-only ablations (wrong results): 1 no transform arithmetic | 2 no transform at all | 16 no U transfers | 32 no raw transfers | 64 no epilogue stores | 128 no operand reads inside the stages
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

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // a 16-byte global access on a dword-aligned address

// position (i, j) of the transformed domain in the fragment order: the column halves j < 3 / j >= 3 as the xi ranges [0, 18) / [18, 36)
static constexpr __host__ __device__ int xi_of(int i, int j) { return j < 3 ? 3 * i + j : 18 + 3 * i + (j - 3); }
// register (dd[6 i + j]) of the value at position xi of the fragment order
static constexpr __host__ __device__ int w4p_dd_of(int xi) { return 6 * ((xi % 18) / 3) + 3 * (xi / 18) + (xi % 18) % 3; }

}   // namespace

// 16-byte / 4-byte global stores written THROUGH to memory at device scope (sc1): what another workgroup of this launch, on any XCD, reads after
// the writer has drained them (s_waitcnt vmcnt(0)) and counted its arrival -- no release fence, the XCD's L2 keeps no dirty line (PROG below)
static __device__ __forceinline__ void store16_sc1(float *p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory"); }
static __device__ __forceinline__ void store4_sc1(float *p, float v) { asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory"); }
// ... and at SYSTEM scope (sc0 sc1): the gather jobs' output rows, which the host pipeline's drainer reads from page-locked host memory as soon as the job's flag
// says so.  Plain stores stayed in the XCD's L2 until the launch ended -- fine-grained host memory is only promised at system-scope synchronisation -- and the
// flag (a system-scope store) arrived long before the rows it announced (measured: stale rows in a third of the frame).
static __device__ __forceinline__ void store16_sys(float *p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory"); }
static __device__ __forceinline__ void store4_sys(float *p, float v) { asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory"); }

// a batch of N 16-byte sc1 loads (scalar base b[k] + one 32-bit lane offset) issued back to back, and the "+v" pins that keep every destination
// behind the batch's wait (an asm load's destination counts as written at the end of its statement: nothing may read or move it before the wait)
template <int K, int N>
static __device__ __forceinline__ void load16_sc1_batch(f32x4 (&t)[N], unsigned voff, const char *const (&b)[N])
{
    if constexpr (K < N) {
        // (the load reads a COPY of its scalar base made by the scalar ALU inside the statement: the base may just have been restored from a spill lane, and a VALU
        //  write of an SGPR needs five wait states before a VMEM instruction reads it as its address -- the hazard recogniser does not look inside an asm statement,
        //  and without protection the load went off a stale base and faulted; VALU -> SALU is interlocked, SALU -> VMEM needs nothing: w2xc_device.h)
        unsigned long long bc;
        asm volatile("s_mov_b64 %1, %3\n\tglobal_load_dwordx4 %0, %2, %1 sc1" : "=v"(t[K]), "=&s"(bc) : "v"(voff), "s"(b[K]) : "memory");
        load16_sc1_batch<K + 1, N>(t, voff, b);
    }
}
template <int K, int N>
static __device__ __forceinline__ void pin_batch(f32x4 (&t)[N])
{
    if constexpr (K < N) {
        asm volatile("" : "+v"(t[K]));
        pin_batch<K + 1, N>(t);
    }
}
template <int K, int N>
static __device__ __forceinline__ void sum_batch(f32x4 &v, const f32x4 (&t)[N])
{
    if constexpr (K < N) {
        v += t[K];
        sum_batch<K + 1, N>(v, t);
    }
}

// One gather job of a PROG launch (below): the last layer's outputs of job (jr, jg) -- rows [16 jr - wino_py - g_off, + 16) x columns [256 jg, + 256) clipped to its
// plane --: out(y, x) = leaky(bias + sum over taps and 64-plane blocks of G[block][tap][y + g_off + ty][x + tx]), summed in conv3x3_last_gather_x4's order (taps outer,
// blocks inner: the two paths are bit-identical).  The other workgroups' tap planes were written through (sc1) and drained before their arrivals were counted: they
// are read with sc1 loads (past this CU's L1), no fence.  Two pixel quads per thread = 36 (NOB = 2) 16-byte loads in flight, issued as one asm batch and waited for once:
// left to the compiler -- which schedules for registers in this kernel -- every load was followed by its own vmcnt(0) (365 us per job instead of ~5).
// A FUNCTION OF ITS OWN, never inlined: as a lambda inside the kernel this code cost layer 6 0.2 ms without ever running (profiles/r6_sweeps.log 1d) -- its scalar
// state competed with the stage loop's for SGPRs and the epilogue came out differently; behind a call it has its own registers.
struct W4ProgJob {
    const float *G; long long ts, gs, rs;          // partial tap planes G[block][tap][y][x]: block / tap-plane / row strides in floats
    float *out; long long out_rs;                  // the last layer's output rows
    const float *bias;
    unsigned *flags; unsigned epoch;               // host pipeline: one word per job, written behind the rows (NULL: none)
    int g_h, g_w, g_off, wino_py, ngroups;
};
template <int NOB>
static __device__ __attribute__((noinline)) void w4_prog_job(W4ProgJob a, int jr, int jg)
{
    constexpr int ROWS = 16, GW = 8;
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };   // (arguments of a device function arrive in VGPRs: everything wave-uniform is made provably so)
    auto uni64 = [&](unsigned long long v) { return ((unsigned long long)(unsigned)uni((int)(unsigned)(v >> 32)) << 32) | (unsigned)uni((int)(unsigned)v); };
    jr = uni(jr); jg = uni(jg);
    const int g_h = uni(a.g_h), g_w = uni(a.g_w), g_off = uni(a.g_off);
    const long long rs = (long long)uni64((unsigned long long)a.rs), gs = (long long)uni64((unsigned long long)a.gs), ts = (long long)uni64((unsigned long long)a.ts);
    const long long out_rs = (long long)uni64((unsigned long long)a.out_rs);
    const float *G = reinterpret_cast<const float *>(uni64((unsigned long long)a.G));
    float *outp = reinterpret_cast<float *>(uni64((unsigned long long)a.out));
    const int tid = (int)threadIdx.x;
    const int y_first = ROWS * jr - uni(a.wino_py) - g_off;
    const int y_lo = y_first > 0 ? y_first : 0, y_hi = y_first + ROWS < g_h ? y_first + ROWS : g_h;
    const int x_lo = jg * GW * 32, x_hi = (jg + 1) * GW * 32 < g_w ? (jg + 1) * GW * 32 : g_w;
    const int nq = x_hi > x_lo ? (x_hi - x_lo + 3) >> 2 : 0;
    const int total = y_hi > y_lo ? (y_hi - y_lo) * nq : 0;   // (jobs of tile rows / groups outside the last layer's plane: nothing to sum, still reported)
    const float b = reinterpret_cast<const float *>(uni64((unsigned long long)a.bias))[0];
    // scalar bases of the 9 NOB (tap, block) planes, the tap's row and column shift included
    const char *gb[9 * NOB];
#pragma unroll
    for (int tap = 0; tap < 9; tap++)
#pragma unroll
        for (int hf = 0; hf < NOB; hf++) gb[tap * NOB + hf] = reinterpret_cast<const char *>(uni64((unsigned long long)(G + hf * ts + tap * gs + (long long)(tap / 3) * rs + (tap % 3))));
    for (int q0 = tid; q0 < total; q0 += 1024) {
        f32x4 t0[9 * NOB], t1[9 * NOB];
        // quad q -> (row, first column); a ragged last quad of a row is loaded from the row's last four columns instead (still inside the row: it has two
        // more) and picked apart below.  32-bit byte offsets inside a tap plane (the launcher checks the plane's size).
        const bool on1 = q0 + 512 < total;
        const int qa = q0, qb = on1 ? q0 + 512 : q0;
        const int ya = qa / nq, yb = qb / nq;
        const int yy0 = y_lo + ya, yy1 = y_lo + yb, xx0 = x_lo + (qa - ya * nq) * 4, xx1 = x_lo + (qb - yb * nq) * 4;
        const bool whole0 = xx0 + 4 <= g_w, whole1 = xx1 + 4 <= g_w;
        const unsigned voff0 = (unsigned)(((long long)(yy0 + g_off) * rs + (whole0 ? xx0 : g_w - 4)) * 4);
        const unsigned voff1 = (unsigned)(((long long)(yy1 + g_off) * rs + (whole1 ? xx1 : g_w - 4)) * 4);
        load16_sc1_batch<0, 9 * NOB>(t0, voff0, gb);
        if (on1) load16_sc1_batch<0, 9 * NOB>(t1, voff1, gb);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // one wait for the batch
        pin_batch<0, 9 * NOB>(t0);
        pin_batch<0, 9 * NOB>(t1);
        auto finish = [&](bool on, const f32x4 (&t)[9 * NOB], int yy, int xx, bool whole) {
            if (!on) return;
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            sum_batch<0, 9 * NOB>(v, t);
            float *oq = outp + (long long)yy * out_rs + xx;
            if (whole) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = leaky(v[e] + b);
                store16_sys(oq, o);
            } else {
                // the row's last 1..3 pixels: the batch held columns g_w - 4 .. g_w - 1 (+ the tap's shift); pixel xx + e sits at index xx + e - (g_w - 4)
                const int sh = xx - (g_w - 4);
                for (int e = 0; xx + e < g_w; e++) {
                    const int idx = sh + e;
                    const float ve = idx == 0 ? v[0] : idx == 1 ? v[1] : idx == 2 ? v[2] : v[3];
                    store4_sys(oq + e, leaky(ve + b));
                }
            }
        };
        finish(true, t0, yy0, xx0, whole0);
        finish(on1, t1, yy1, xx1, whole1);
    }
    // flags (host pipeline): the job's rows are on their way to host memory (`out` is page-locked host memory there: the stores are posted PCIe writes, issued at
    // system scope): every wave waits for its stores to have left, then ONE system-scope store publishes the job to the drainer thread, behind the data on the same link
    unsigned *flags = reinterpret_cast<unsigned *>(uni64((unsigned long long)a.flags));
    if (flags) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tid == 0) __hip_atomic_store(flags + (jr * uni(a.ngroups) + jg), (unsigned)uni((int)a.epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// PROG (with FUSE7): the launch FINISHES the fused last layer itself and finishes it in row order, so that the output plane completes top to bottom
// while the launch is still running (round 6; convertRoutine.cpp:143-161's stitch no longer waits for the layer's end):
//   * schedule: XCD k owns the tile columns [(k tiles_x + phi) / 8, ((k + 1) tiles_x + phi) / 8) of tile row r, phi = r & 7, and walks them row by row --
//     all eight XCDs work on the same tile rows at the same time (the strip walk above finishes the lower 40 % of every strip only at the very end);
//     the band edges move by at most one tile from row to row and eight rows hold exactly tiles_x tiles of every XCD: equal shares for any width;
//   * every item writes its partial tap planes through to memory (sc1) and, one item later -- when the stores have long drained --, counts its
//     arrival on the GATHER JOBS it feeds: job (r, g) = the last layer's outputs whose first tap row lies in tile row r and whose columns lie in the
//     tile group [8 g, 8 g + 8); it needs the tiles of rows r, r + 1 and columns 8 g .. 8 g + 8 (two more tap rows below, two more columns to the right);
//   * the workgroup whose arrival completes a job runs it: conv3x3_last_gather's sum (taps outer, 64-plane blocks inner: bit-identical), bias,
//     LeakyReLU, stores into the output plane; with d.prog_flags the job's completion is published to the host (system scope) for the drainer.
//   No workgroup ever waits for another one (no spinning, no residency assumption): arrivals are returning atomics, the last one does the work.
template <int CIN, int COUT, bool OUT_PLANAR, bool IN_NHWC = false, bool FUSE7 = false, bool PROG = false>
__global__ void __launch_bounds__(512, 2) conv3x3_wino4(W2xcConvDesc d, int tiles_x, int nitems)
{
    static_assert(!PROG || FUSE7, "PROG finishes the fused last layer");
    constexpr int ROWS = 16;
    constexpr int NST = CIN / 4;                            // stages (4-channel slices) per item
    constexpr int NOB = COUT / 64;                          // 64-plane blocks
    constexpr int CHS = 168;                                // chunks per channel of a raw buffer (18 rows x 9 quads = 162, + 6: stride = 8 mod 16)
    constexpr int RAW_PIECES = 11;                          // 4 x 168 = 672 chunks = 10.5 pieces of 64 x 16 bytes
    constexpr unsigned RAW_BYTES = RAW_PIECES * 1024;
    constexpr unsigned U_BASE = 3 * RAW_BYTES, U_BYTES = 36 * 1024;
    constexpr unsigned V_BASE = U_BASE + 2 * U_BYTES, V_BYTES = 18 * 1024;
    constexpr unsigned SPARE_BASE = V_BASE + 2 * V_BYTES;   // 18 KiB: with U slot 1 the parking area across an epilogue
    constexpr unsigned BIAS_BASE = SPARE_BASE + V_BYTES;
    static_assert(CIN % 16 == 0 && COUT % 64 == 0 && NST % 4 == 0 && NST >= 8, "planes");
    constexpr int STRIP = 16;
    constexpr int GW = 8;                                   // PROG: tiles per gather-job column group
    constexpr unsigned TRIG_BASE = BIAS_BASE + COUT * 4;    // PROG: one LDS word, the jobs this workgroup's arrivals completed
    const int tiles_y = nitems / (NOB * tiles_x);
    const int pxcd = blockIdx.x & 7;
    auto band_lo = [&](int k, int r) { return (k * tiles_x + (r & 7)) >> 3; };   // PROG: first tile column of XCD k in tile row r
    auto tile_coords = [&](int pt_, int &ty_, int &tx_) {     // strips of 16 tiles, row by row inside a strip (the next round of an XCD is the tile row below)
        if constexpr (PROG) {   // pt_ = index into THIS XCD's band, row by row; a period of eight rows holds exactly tiles_x of its tiles
            const int p8 = pt_ / tiles_x;
            int rem = pt_ - p8 * tiles_x, j = 0, lo = band_lo(pxcd, 0), wd = band_lo(pxcd + 1, 0) - lo;
            while (rem >= wd && j < 7) {
                rem -= wd;
                j++;
                lo = band_lo(pxcd, j);
                wd = band_lo(pxcd + 1, j) - lo;
            }
            ty_ = 8 * p8 + j;
            tx_ = lo + rem;
            return;
        }
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
    // ONE lane-linear VGPR lives through the kernel (lane * 16); everything else that depends on the lane is derived from an opaque copy of it where it is
    // used -- hoisted out of the item loop, the ~30 lane-derived values of the epilogue, the transfers and the transform overflow the 256 registers
    // into scratch, and a scratch reload inside a stage waits with vmcnt(0) for every transfer in flight
    const unsigned b_voff = (unsigned)lane * 16u;
    auto lin = [&]() { unsigned v = b_voff; asm volatile("" : "+v"(v)); return v; };
    auto lane_o = [&]() { return (int)(lin() >> 4); };

    const int xcd = blockIdx.x & 7, per = gridDim.x >> 3;
    const int cq = nitems >> 3, cr = nitems & 7;
    int chunk_begin = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
    int chunk_end = chunk_begin + cq + (xcd < cr ? 1 : 0);
    if constexpr (PROG) {   // items = indices into this XCD's own band (tile_coords above)
        int nloc = (tiles_y >> 3) * tiles_x;
        for (int j = 0; j < (tiles_y & 7); j++) nloc += band_lo(xcd + 1, j) - band_lo(xcd, j);
        chunk_begin = 0;
        chunk_end = nloc * NOB;
    }
#ifndef W4_PAIR
#define W4_PAIR 0   // (experiment, profiles/r6_sweeps.log 3) 1: a workgroup runs BOTH 64-plane items of a tile back to back instead of the two items side by side on two workgroups
#endif
    int item0 = chunk_begin + (blockIdx.x >> 3);
    int nmy = (chunk_end - item0 + per - 1) / per;
    if constexpr (W4_PAIR && NOB == 2 && !PROG) {
        const int ntile = nitems >> 1, tq = ntile >> 3, tr = ntile & 7;
        const int tb = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, te = tb + tq + (xcd < tr ? 1 : 0);
        item0 = tb + (blockIdx.x >> 3);            // (a TILE index in this form)
        if (item0 >= te) return;
        nmy = 2 * ((te - item0 + per - 1) / per);
    } else if (item0 >= chunk_end) return;
    auto item_of = [&](int n) {
        const int nn = n < nmy ? n : nmy - 1;
        if constexpr (W4_PAIR && NOB == 2 && !PROG) return 2 * (item0 + (nn >> 1) * per) + (nn & 1);
        else return item0 + nn * per;
    };

    for (int c = threadIdx.x; c < COUT; c += 512) lds[BIAS_BASE / 4 + c] = d.bias[c];   // (visible after the prologue barrier)
    if constexpr (PROG) {
        if (threadIdx.x < 4) lds[TRIG_BASE / 4 + threadIdx.x] = 0.0f;   // no job to run, no ticket in hand
    }

    // ---- raw tile transfers: chunk ci = piece * 64 + lane -> (channel kk, row R, quad q); wave w < DW sends pieces w, w + DW, ... ----
    const long long cs4 = d.in_cs * 4, rs4 = d.in_rs * 4;     // bytes
    constexpr int DW = W4_DMA4 ? 4 : 8;                        // waves that issue transfers
    constexpr int RJ = (RAW_PIECES + DW - 1) / DW;             // raw pieces per such wave (the last one: waves < RAW_PIECES - (RJ - 1) * DW = 3 only)
    constexpr int UQ = 36 / DW + (36 % DW ? 1 : 0);            // U pieces per such wave (8 waves: the fifth from waves < 4 only)
    unsigned voff[RJ];
    const char *a_base;
    int xlim_r;                                                // in_w - x0 of the tile the raw cursor is in (patch columns >= it are outside the plane)
    auto tile_offsets = [&](int it) {
        int ty_, tx_;
        tile_coords(it / NOB, ty_, tx_);
        const int lane_t = lane_o();   // (nothing of this hoisted out of the item loop)
        const int y0 = ty_ * ROWS - d.wino_py + d.off_y, x0 = tx_ * 32 + d.off_x;
        const int yb = clampi(y0, 0, d.in_h - 1);
        a_base = reinterpret_cast<const char *>(d.in) + (long long)yb * rs4;
        xlim_r = d.in_w - x0;
        if constexpr (IN_NHWC) {
            // NHWC input (32 planes: a pixel = one 128-byte line, written by conv3x3_first / conv3x3_wino): a chunk = channels 4 s .. 4 s + 3 of ONE pixel,
            // chunk index R * 36 + (x % 4) * 9 + x / 4 -- the pixels of a row grouped by column mod 4, so that the eight block columns of a patch
            // position are consecutive chunks (conflict-free ds_read_b32 of a lane's channel); columns clamped (replicate) like the rows
            xlim_r = 64;   // (nothing to mask)
#pragma unroll
            for (int jj = 0; jj < RJ; jj++) {
                int ci = (jj * DW + wave) * 64 + lane_t;
                ci = ci < (ROWS + 2) * 36 ? ci : (ROWS + 2) * 36 - 1;
                const int R = ci / 36, slot = ci - R * 36;
                const int x = 4 * (slot % 9) + slot / 9;
                const int gy = clampi(y0 + R, 0, d.in_h - 1) - yb;
                const int gx = clampi(x0 + x, 0, d.in_w - 1);
                voff[jj] = (unsigned)((long long)gy * rs4 + (long long)gx * (CIN * 4));
            }
            return;
        }
        const int xq_last = (d.in_w - 1) & ~3;
#pragma unroll
        for (int jj = 0; jj < RJ; jj++) {
            const int ci = (jj * DW + wave) * 64 + lane_t;
            int kk = ci / CHS;
            kk = kk < 4 ? kk : 3;
            const int rem = ci - kk * CHS;
            int R = rem / 9;
            const int q = rem - R * 9;
            R = R < ROWS + 2 ? R : ROWS + 1;
            const int gy = clampi(y0 + R, 0, d.in_h - 1) - yb;
            int gx = x0 + 4 * q;
            gx = gx < xq_last ? gx : xq_last;
            voff[jj] = (unsigned)((long long)kk * cs4 + (long long)gy * rs4 + (long long)gx * 4);
        }
    };
    // LDS destinations of the transfers are (wave base + immediate): as precomputed wave-uniform values the ~50 of them are hoisted out of the loops
    // and the SGPR file overflows into VGPR lanes and scratch
    const unsigned wbase = (unsigned)__builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);
    // raw piece jj * 8 + wave of 4-channel slice `slice` of the tile a_base / voff describe, into the raw buffer at byte offset roff
    auto dma_raw = [&](auto JJ, unsigned roff, int slice) {
        constexpr int jj = decltype(JJ)::value;
        const char *sbase = a_base + (long long)slice * (IN_NHWC ? 16 : 4 * cs4);
        lds_dma16_si<jj * DW * 1024u>(sbase, voff[jj], wbase + roff);
    };
    // U of (64-plane block ob, stage s_): 36 pieces of 1 KiB (one per xi); wave w < DW sends xi = w, w + DW, ...
    const char *wpk_w = reinterpret_cast<const char *>(d.wpk) + (size_t)wave * 1024;
    auto dma_u = [&](int ob, int s_, auto SLOT, auto Q) {     // piece xi = DW q + wave
        constexpr unsigned slot = decltype(SLOT)::value;
        constexpr int q = decltype(Q)::value;
        const char *sbase = wpk_w + ((size_t)(ob * NST + s_) * 36 + q * DW) * 1024;
        lds_dma16_si<U_BASE + slot * U_BYTES + q * DW * 1024u>(sbase, b_voff, wbase);
    };
#ifndef W4_DMA_BASES
#define W4_DMA_BASES 1   // 1: a stage's U pieces go off scalar bases formed ONCE per stage / per pair of pieces (immediate offsets -4096 | 0), 0: every piece forms its own
#endif
    // (one instantiation -- 32 planar planes in, 128 NHWC planes out, not on the default path -- spills four registers with the two more live scalars: it keeps the old form)
    constexpr bool DMAB = W4_DMA_BASES && W4_DMA4 && !(CIN == 32 && COUT == 128 && !OUT_PLANAR && !IN_NHWC && !FUSE7);
    // the same off a base the caller keeps: u_pair = (this wave's piece q | 1 of the stage), pieces q even at -4096, q odd at 0
    auto dma_u_pair = [&](const char *u_pair, auto SLOT, auto Q) {
        constexpr unsigned slot = decltype(SLOT)::value;
        constexpr int q = decltype(Q)::value;
        lds_dma16_sio<U_BASE + slot * U_BYTES + q * DW * 1024u, (q & 1) ? 0 : -(DW * 1024)>(u_pair, b_voff, wbase);
    };

    // ---- addressing ----
    // MFMA operands: lane-linear 16-byte quads, (wave-uniform base) + lane * 16: A = U at U_BASE + slot * U_BYTES + pt * 1024 + (xi / 4) * 4096,
    // B = V at V_BASE + slot * V_BYTES + bt * 1024 + (xi / 4) * 2048.  ONE lane-linear VGPR (b_voff) + SGPR bases: the copies the compiler would
    // otherwise keep per base cost registers this kernel does not have.
    const unsigned ua_u = U_BASE + (unsigned)pt * 1024u, va_u = V_BASE + (unsigned)bt * 1024u;
#ifndef W4_KEEP_BASES
#define W4_KEEP_BASES 1   // 1: the operand base addresses (U slot 0 / 1, V) live in three registers of their own; 0: rebuilt from the lane register where used
#endif
    // The operand bases of a stage: with W4_KEEP_BASES they are three more lane-linear registers that live through the kernel (a ds_read's 16-bit immediate
    // reaches both V slots from one base, U's two slots need one each); rebuilt per use they were 7 VALU instructions per wave and stage -- and a VALU
    // instruction is 4 cycles of a SIMD that issues nothing else meanwhile
    unsigned ua0_v = b_voff + ua_u, ua1_v = b_voff + ua_u + U_BYTES, va_v = b_voff + va_u;
    asm volatile("" : "+v"(ua0_v), "+v"(ua1_v), "+v"(va_v));
    auto ua_of = [&](unsigned slot) { return W4_KEEP_BASES ? ldsb + (slot ? ua1_v : ua0_v) : ldsb + (lin() + (ua_u + slot * U_BYTES)); };
    auto va_of = [&](unsigned slot) { return W4_KEEP_BASES ? ldsb + va_v + slot * V_BYTES : ldsb + (lin() + (va_u + slot * V_BYTES)); };
    // transformer lane (r, kk, c) = block (block row 2 bt + r, column c), channel kk: patch row i, columns 0..3 = chunk (kk, 4 (2 bt + r) + i, c),
    // columns 4, 5 = the first half of chunk (kk, same row, c + 1)
#ifndef W4_KEEP_TR
#define W4_KEEP_TR 1   // 1: the transform's patch-read and V-write lane offsets live in two registers of their own (0: rebuilt from the lane register per quarter)
#endif
    auto tr_rd_calc = [&]() {
        const int l = lane_o(), tr_r = l >> 5, tr_k = (l >> 3) & 3, tr_c = l & 7;
        return IN_NHWC ? (unsigned)((4 * (2 * bt + tr_r) * 36 + tr_c) * 16 + tr_k * 4)                // + buffer + (i * 36 + (j & 3) * 9 + (j >> 2)) * 16
                       : (unsigned)((tr_k * CHS + 4 * (2 * bt + tr_r) * 9 + tr_c) * 16);        // + buffer + i * 144 (+ 16)
    };
    // V-slot address of this lane's patch in the fragment order lane = 16 kk + block (+ slot * V_BYTES + (xi / 4) * 2048)
    auto tr_wr_calc = [&]() {
        const int l = lane_o(), tr_r = l >> 5, tr_k = (l >> 3) & 3, tr_c = l & 7;
        return V_BASE + (unsigned)bt * 1024u + (unsigned)((tr_k * 16 + tr_r * 8 + tr_c) * 16);
    };
    unsigned tr_rd_v = tr_rd_calc(), tr_wr_v = tr_wr_calc();
    asm volatile("" : "+v"(tr_rd_v), "+v"(tr_wr_v));
    auto tr_rd = [&]() { return W4_KEEP_TR ? tr_rd_v : tr_rd_calc(); };
    auto tr_wr = [&]() { return W4_KEEP_TR ? tr_wr_v : tr_wr_calc(); };

    // The input transform V = B^T d B of a patch set (16 blocks x 4 channels = one patch per lane) by ONE wave in FOUR QUARTERS over four consecutive
    // stages, the 36 values in REGISTERS in between:
    //   Q0 (stage n - 4)  the raw patch (6 x (ds_read_b128 + ds_read_b64)), columns outside the plane zeroed, the row pass d B of rows 0..2 (3 x 14 fma / add)
    //   Q1 (stage n - 3)  the row pass of rows 3..5
    //   Q2 (stage n - 2)  the column pass B^T (.) of columns 0..2
    //   Q3 (stage n - 1)  the column pass of columns 3..5, V of stage n as nine ds_write_b128 into the slot stage n - 2 has read
    // With PH = (pt - 2 bt) mod 4 a wave runs quarter (stage - PH) mod 4: it transforms the slices n = PH (mod 4) of its block tile, every wave carries
    // the same 42 VALU instructions in every stage, and the two waves of a SIMD (same pt) are two quarters apart (raw reads beside pure arithmetic).
    // The pipeline runs ACROSS items (the last four stages of an item work on the first slices of the next): the three waves of a block tile that
    // are in mid-transform at an item's end park their 36 values in LDS across the epilogue (whose registers are full) -- in the U slot and the 18 KiB
    // that are idle then -- and take them back behind it.  Nothing else of a transform touches LDS between its raw reads and its V writes: round 4's
    // earlier forms parked the intermediates of EVERY transform in the V ring (row-pass waves -> column-pass waves), and that traffic alone cost 0.7
    // of layer 6's 6.7 ms (timing-only ablation, profiles/r4_sweeps.log 8).
    // ---- PROG: arrivals and gather jobs (see the comment above the kernel) ----
    const int ngroups = (tiles_x + GW - 1) / GW;
    // lane l < 4 of an item at tile (r, tx): the job it feeds -- (r - (l & 1), tx / 8 - (l >> 1)); the left neighbour group only from the group's first column
    auto job_of_lane = [&](int item, int l, int &jr, int &jg) {
        int r, tx;
        tile_coords(item / NOB, r, tx);
        jr = r - (l & 1);
        jg = tx / GW - (l >> 1);
        return l < 4 && jr >= 0 && ((l >> 1) == 0 || ((tx & (GW - 1)) == 0 && tx > 0));
    };
    auto job_target = [&](int jr, int jg) {   // arrivals that complete job (jr, jg)
        const int rows = jr + 1 < tiles_y ? 2 : 1;
        const int c1 = (jg + 1) * GW + 1;
        return NOB * rows * ((c1 < tiles_x ? c1 : tiles_x) - jg * GW);
    };
    // The counter buffer (zeroed by the launcher in front of every launch): [0, njobs) arrivals per job | [njobs, 2 njobs) the READY QUEUE, slot -> job + 1 |
    // head, tail.  The workgroup whose arrival completes a job only PUSHES it (tail++, then the slot); any workgroup that sees head < tail at an item
    // boundary draws a ticket (head++) and runs the job of that slot at its next boundary.  (Letting the last arriver run the job itself fed back on
    // itself: the slowest workgroup of a neighbourhood is the last arriver every round, got all its jobs -- +20 % on the launch, measured.)
    // All of it is done by wave 7, which issues neither tap-plane stores nor transfers: the compiler's wait for a returned value is vmcnt(0), and in a
    // storing wave that waits for the stores just issued as well.
    const int njobs = tiles_y * ngroups;
    unsigned *const q_slots = d.prog_cnt + njobs, *const q_head = d.prog_cnt + 2 * njobs, *const q_tail = q_head + 1;
    unsigned *const lds_action = reinterpret_cast<unsigned *>(ldsb + TRIG_BASE), *const lds_ticket = lds_action + 1;   // job + 1 to run now | slot + 1 drawn, not yet run
    constexpr int CW = 7;   // the control wave
    // control wave, early in the epilogue: count the arrival of `item` (whose tap planes have drained) on its jobs -- every lane keeps what its job's
    // counter held before --, and look at the queue: the slot of the ticket in hand, or head and tail
    auto prog_early = [&](int item, unsigned &tick, unsigned &q0, unsigned &q1) {
        tick = 0xFFFFFFFFu;
        if (item >= 0) {
            int jr, jg;
            if (job_of_lane(item, lane_o(), jr, jg)) tick = __hip_atomic_fetch_add(d.prog_cnt + (jr * ngroups + jg), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned t = *lds_ticket;
        q0 = __hip_atomic_load(t ? q_slots + (t - 1) : q_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        q1 = __hip_atomic_load(q_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // control wave, late in the epilogue: push the jobs the arrivals completed; take the job of the ticket in hand if its slot has been filled, or draw a
    // ticket if a ready job has none -> the LDS words every wave reads behind the next barrier
    auto prog_late = [&](int item, unsigned tick, unsigned q0, unsigned q1) {
        if (item >= 0) {
            int jr, jg;
            const bool valid = job_of_lane(item, lane_o(), jr, jg);
            if (valid && tick == (unsigned)(job_target(jr, jg) - 1)) {   // this arrival was the job's last one
                const unsigned slot = __hip_atomic_fetch_add(q_tail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(q_slots + slot, (unsigned)(jr * ngroups + jg) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        unsigned t = *lds_ticket, action = 0;
        if (t) {
            if (q0) { action = q0; t = 0; }
        } else if ((int)(q1 - q0) > 0) {
            unsigned h = 0;
            if (lane_o() == 0) h = __hip_atomic_fetch_add(q_head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            h = (unsigned)__builtin_amdgcn_readfirstlane((int)h);
            t = h < (unsigned)njobs ? h + 1 : 0;   // (a ticket past the last job: every job has been drawn)
        }
        if (lane_o() == 0) { *lds_action = action; *lds_ticket = t; }
    };
    // every wave: run the job the control wave has named
    auto prog_run = [&]() {
        const unsigned action = (unsigned)__builtin_amdgcn_readfirstlane((int)*lds_action);
        if (action) {
            W4ProgJob a;
            a.G = d.out; a.ts = d.out_ts; a.gs = d.out_gs; a.rs = d.out_rs;
            a.out = d.g_out; a.out_rs = d.g_out_rs; a.bias = d.g_bias; a.flags = d.prog_flags; a.epoch = d.prog_epoch;
            a.g_h = d.g_h; a.g_w = d.g_w; a.g_off = d.g_off; a.wino_py = d.wino_py; a.ngroups = ngroups;
            w4_prog_job<NOB>(a, (int)(action - 1) / ngroups, (int)(action - 1) % ngroups);
        }
        return action;
    };

    float dd[36];
    auto raw_read = [&](const char *src, auto I_) {          // patch row i
        constexpr int i = decltype(I_)::value;
        if constexpr ((W4_ABL & 2) != 0) {
            static_for<0, 6>([&](auto JJ) { dd[i * 6 + decltype(JJ)::value] = (float)(i + decltype(JJ)::value); });
        } else if constexpr (IN_NHWC) {
            static_for<0, 6>([&](auto JJ) {
                constexpr int j = decltype(JJ)::value;
                dd[i * 6 + j] = *reinterpret_cast<const float *>(src + (i * 36 + (j & 3) * 9 + (j >> 2)) * 16);
            });
        } else {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(src + i * 144);
            const f32x2 b = *reinterpret_cast<const f32x2 *>(src + i * 144 + 16);
            dd[i * 6 + 0] = a[0]; dd[i * 6 + 1] = a[1]; dd[i * 6 + 2] = a[2]; dd[i * 6 + 3] = a[3];
            dd[i * 6 + 4] = b[0]; dd[i * 6 + 5] = b[1];
        }
    };
    auto raw_mask = [&](int xlim) {                           // patch columns outside the plane: zero (wave-uniform test first)
        if (xlim < 34) {
            const int lim = xlim - 4 * (lane_o() & 7);
            static_for<0, 36>([&](auto E) {
                constexpr int e = decltype(E)::value;
                dd[e] = (e % 6) < lim ? dd[e] : 0.0f;
            });
        }
    };
    // (the pins keep a pass in the stage it was written in: without them the compiler sinks it into the stage that consumes its results)
    auto row_pass = [&](auto I_) {                            // s[i][.] = d[i][.] B
        constexpr int i = decltype(I_)::value;
        if constexpr (!(W4_ABL & 3)) bt6(dd[i * 6 + 0], dd[i * 6 + 1], dd[i * 6 + 2], dd[i * 6 + 3], dd[i * 6 + 4], dd[i * 6 + 5]);
        if constexpr (!(W4_ABL & 3)) asm volatile("" : "+v"(dd[i * 6 + 0]), "+v"(dd[i * 6 + 1]), "+v"(dd[i * 6 + 2]), "+v"(dd[i * 6 + 3]), "+v"(dd[i * 6 + 4]), "+v"(dd[i * 6 + 5]));
    };
    auto col_pass = [&](auto J_) {                            // V[.][j] = B^T s[.][j]
        constexpr int j = decltype(J_)::value;
        if constexpr (!(W4_ABL & 3)) bt6(dd[0 * 6 + j], dd[1 * 6 + j], dd[2 * 6 + j], dd[3 * 6 + j], dd[4 * 6 + j], dd[5 * 6 + j]);
        if constexpr (!(W4_ABL & 3)) asm volatile("" : "+v"(dd[0 * 6 + j]), "+v"(dd[1 * 6 + j]), "+v"(dd[2 * 6 + j]), "+v"(dd[3 * 6 + j]), "+v"(dd[4 * 6 + j]), "+v"(dd[5 * 6 + j]));
    };
    // quad q of the fragment order = positions xi = 4 q .. 4 q + 3; position (i, j) sits at xi_of(i, j), its value in dd[6 i + j]
    auto v_write = [&](char *dst, auto Q_) {
        constexpr int q = decltype(Q_)::value;
        if constexpr (!(W4_ABL & 2))
        {
#ifdef W4_VW32
            static_for<0, 4>([&](auto E_) { constexpr int e = decltype(E_)::value; *reinterpret_cast<float *>(dst + q * 2048 + e * 4) = dd[w4p_dd_of(4 * q + e)]; });
#else
            *reinterpret_cast<f32x4 *>(dst + q * 2048) = f32x4{dd[w4p_dd_of(4 * q)], dd[w4p_dd_of(4 * q + 1)], dd[w4p_dd_of(4 * q + 2)], dd[w4p_dd_of(4 * q + 3)]};
#endif
        }
    };

    // A run-time "which quarter now" around the stage bodies joins 144 accumulators in phi nodes and the register allocator gives up: the item loop exists
    // four times (one copy per PH, unrolled by four stages with the quarters fixed at compile time) and a wave picks its copy once.
    auto run = [&](auto PH_) {
    constexpr int PH = decltype(PH_)::value;
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using U0 = std::integral_constant<unsigned, 0u>;
    // where this wave parks across an epilogue: six parkers (three per block tile) x 9 KiB = U slot 1 (idle behind the last stage's closing barrier) + the spare 18 KiB
    const unsigned park_u = (bt * 3 + PH - 1) < 4 ? U_BASE + U_BYTES + (unsigned)(bt * 3 + PH - 1) * 9216u : SPARE_BASE + (unsigned)(bt * 3 + PH - 1 - 4) * 9216u;
    auto park = [&]() {
        char *pa = ldsb + (lin() + park_u);
        static_for<0, 9>([&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
            *reinterpret_cast<f32x4 *>(pa + q * 1024) = f32x4{dd[4 * q], dd[4 * q + 1], dd[4 * q + 2], dd[4 * q + 3]};
        });
    };
    auto unpark = [&]() {
        const char *pa = ldsb + (lin() + park_u);
        static_for<0, 9>([&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(pa + q * 1024);
            dd[4 * q] = v[0]; dd[4 * q + 1] = v[1]; dd[4 * q + 2] = v[2]; dd[4 * q + 3] = v[3];
        });
    };
    // a whole quarter outside the stages (kernel prologue)
    auto quarter = [&](auto Q_, const char *src, char *dst, int xlim) {
        constexpr int q = decltype(Q_)::value;
        if constexpr (q == 0) {
            static_for<0, 6>([&](auto I) { raw_read(src, I); });
            raw_mask(xlim);
        }
        if constexpr (q < 2) static_for<3 * q, 3 * q + 3>([&](auto I) { row_pass(I); });
        else static_for<3 * (q - 2), 3 * (q - 2) + 3>([&](auto J) { col_pass(J); });
        if constexpr (q == 3) static_for<0, 9>([&](auto Q) { v_write(dst, Q); });
    };
    // ---- kernel prologue: raw slices 0..2 and U(stage 0) of the first item; the quarters that precede stage 0 (slice 0 whole -> V slot 0, slice 1 Q0..Q2,
    //      slice 2 Q0 Q1); raw slices 3..5; slice 3 Q0 ----
    tile_offsets(item_of(0));
    int xlim_cur = xlim_r;                                      // in_w - x0 of the current item's tile
    auto raw_all = [&](unsigned roff, int slice) {              // this wave's pieces of a slice
        if (wave < DW) {
            static_for<0, RJ - 1>([&](auto JJ) { dma_raw(JJ, roff, slice); });
            if (wave < RAW_PIECES - (RJ - 1) * DW) dma_raw(std::integral_constant<int, RJ - 1>{}, roff, slice);
        }
    };
    for (int sl = 0; sl < 3; sl++) raw_all((unsigned)sl * RAW_BYTES, sl);
    if (wave < DW) {
        const int ob0 = item_of(0) % NOB;
        static_for<0, UQ - 1>([&](auto Q) { dma_u(ob0, 0, U0{}, Q); });
        if (wave < 36 - (UQ - 1) * DW) dma_u(ob0, 0, U0{}, std::integral_constant<int, UQ - 1>{});
    }
    W2XC_WAIT_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (PH < 3) {
        const char *src = ldsb + PH * RAW_BYTES + tr_rd();
        static_for<0, 4 - PH>([&](auto Q) { quarter(Q, src, ldsb + tr_wr(), xlim_cur); });
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int sl = 3; sl < 6; sl++) raw_all((unsigned)(sl - 3) * RAW_BYTES, sl);
    W2XC_WAIT_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (PH == 3) quarter(C0{}, ldsb + tr_rd(), nullptr, xlim_cur);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // MFMA operand quads (four xi per ds_read_b128 and operand), read W4_PF groups of four MFMAs ahead.  The closing wait and the barrier of a stage sit in
    // front of its LAST group: behind the barrier the wave reads the first group(s) of the NEXT stage and still has four MFMAs of this one to issue
    // while they arrive -- with the barrier behind the last MFMA every stage began with an exposed LDS round trip (~400 of ~3000 cycles, s_memtime).
    constexpr int PF = W4_PF;
    static_assert(PF == 1 || PF == 2, "operand look-ahead");
    f32x4 a4[PF + 1], b4[PF + 1];
    // register buffer of operand group g of a stage of parity par: g mod 3 with two groups of look-ahead; with one, two buffers whose roles swap with the
    // stage's parity (group 8 and the next stage's group 0 are alive together)
    auto load_first = [&](unsigned slot) {
        const char *ua = ua_of(slot);
        const char *va = va_of(slot);
        static_for<0, PF>([&](auto G) {
            constexpr int g = decltype(G)::value;
            const int b = PF == 2 ? g : (int)slot;
            a4[b] = *reinterpret_cast<const f32x4 *>(ua + g * 4096);
            b4[b] = *reinterpret_cast<const f32x4 *>(va + g * 2048);
        });
    };
    unsigned rd = RAW_BYTES, mid = 2 * RAW_BYTES, fr = 0;   // raw buffers of slice s + 4 (Q0 reads it), s + 5, and the one Q0 of the last stage has read (this stage's transfer: s + 6)
    int stamp = 0;
    (void)stamp;
    W4_STAMP(stamp++);
    for (int n = 0; n < nmy; n++) {
        const int item = item_of(n), item_n = item_of(n + 1);
        f32x4 acc[36];   // (first written by the item's first stage: its MFMAs take 0 as their accumulator input -- 144 v_mov per wave and item less)

        // one stage; J = global stage count mod 4 (NST is a multiple of 4: = s mod 4)
        auto stage = [&](auto J_, auto FIRST_, int s) {
            constexpr int J = decltype(J_)::value;
            constexpr bool FIRST = decltype(FIRST_)::value;   // the item's first stage
            constexpr int QT = (J - PH) & 3;                                              // this wave's quarter, of slice s + 4 - QT
            constexpr unsigned par = J & 1, nxt = par ^ 1u;                               // U / V slot of this stage and of the next
            int u_ob = item % NOB, u_s = s + 1;
            if (s == NST - 1) { u_ob = item_n % NOB; u_s = 0; }
            const char *ua = ua_of(par);
            const char *va = va_of(par);
            // this wave's first U piece of the stage it transfers, formed once (left to the compiler every piece recomputed it from (u_ob, u_s): ~6 scalar
            // instructions per piece in front of an MFMA that waits for them in program order)
            const char *u_st = wpk_w + (size_t)(u_ob * NST + u_s) * 36864, *u_pair = u_st;
            if constexpr (DMAB) asm volatile("" : "+s"(u_st));
            const char *srcQ = nullptr;
            char *dstQ = nullptr;
            if constexpr (QT == 0) srcQ = ldsb + (rd + tr_rd());
            if constexpr (QT == 3) dstQ = ldsb + (tr_wr() + nxt * V_BYTES);
            const int xlimQ = s + 4 < NST ? xlim_cur : xlim_r;
            if constexpr (J == 2) {
                if (s == NST - 6) tile_offsets(item_n);   // from this stage on the raw cursor (six slices ahead) is in the next item's tile
            }
            const int r_slice = s + 6 < NST ? s + 6 : s + 6 - NST;
            static_for<0, 36>([&](auto XI) {
                constexpr int xi = decltype(XI)::value;
                constexpr int g = xi >> 2;
                if constexpr (xi == 32) {
                    // ---- the stage's close, in front of its last four MFMAs ----
                    W4_STAMP(stamp++);
                    // U of the next stage and the raw slice issued one stage ago (Q0 of the NEXT stage reads it) have landed; this stage's raw pieces --
                    // the youngest transfers -- may still fly
                    if constexpr ((W4_ABL & 32) != 0) W2XC_WAIT_VMCNT(0);
                    else if (wave < RAW_PIECES - (RJ - 1) * DW) wait_vmcnt_n(RJ);
                    else if (wave < DW) wait_vmcnt_n(RJ - 1);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    W4_STAMP(stamp++);
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    { const unsigned tr = rd; rd = mid; mid = fr; fr = tr; }
                    if (s != NST - 1) {                     // (an item's last stage: the epilogue comes first, the item loop reads them)
                        if constexpr (PF == 2) load_first(nxt);
                        else {
                            a4[nxt] = *reinterpret_cast<const f32x4 *>(ua_of(nxt));
                            b4[nxt] = *reinterpret_cast<const f32x4 *>(va_of(nxt));
                        }
                    }
                    W4_STAMP(stamp++);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr ((xi & 3) == 0 && g + PF < 9 && !(W4_ABL & 128)) {
                    constexpr int b = PF == 2 ? (g + PF) % 3 : ((g + 1 + (int)par) & 1);
                    a4[b] = *reinterpret_cast<const f32x4 *>(ua + (g + PF) * 4096);
                    b4[b] = *reinterpret_cast<const f32x4 *>(va + (g + PF) * 2048);
                    __builtin_amdgcn_sched_barrier(0);
                }
                {
                    constexpr int b = PF == 2 ? g % 3 : ((g + (int)par) & 1);
                    // (as an instruction with the accumulator tied: left to the register allocator, most of these MFMAs get a destination other than
                    // their accumulator input, the 144 accumulators migrate through the file and some are spilled inside the stages)
                    if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=v"(acc[xi]) : "v"(a4[b][xi & 3]), "v"(b4[b][xi & 3]));
                    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[xi]) : "v"(a4[b][xi & 3]), "v"(b4[b][xi & 3]));
                }
                __builtin_amdgcn_sched_barrier(0);
                // the stage's other work, behind the first MFMA slots: transfers (U pieces first, the raw pieces last: the closing wait leaves them in
                // flight), then this wave's quarter of the input transform
                constexpr int TSTEP = W4_DMA4 ? 1 : 2, tk = xi >= 1 && (xi - 1) % TSTEP == 0 ? (xi - 1) / TSTEP : -1;   // transfer slot behind MFMA xi
                if constexpr (tk >= 0 && tk < UQ && !(W4_ABL & 16)) {
                    constexpr int q = tk;
                    if constexpr (DMAB) {
                        if (wave < (q == UQ - 1 ? 36 - (UQ - 1) * DW : DW)) {
                            if constexpr ((q & 1) == 0) {   // the base of pieces q and q + 1: one 64-bit scalar addition per pair
                                u_pair = u_st + (q + 1) * (DW * 1024);
                                asm volatile("" : "+s"(u_pair));
                            }
                            dma_u_pair(u_pair, std::integral_constant<unsigned, nxt>{}, std::integral_constant<int, q>{});
                        }
                    } else {
                        if (wave < (q == UQ - 1 ? 36 - (UQ - 1) * DW : DW)) dma_u(u_ob, u_s, std::integral_constant<unsigned, nxt>{}, std::integral_constant<int, q>{});
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (tk >= UQ && tk < UQ + RJ && !(W4_ABL & 32)) {
                    constexpr int jj = tk - UQ;      // slice s + 6 into the buffer Q0 of stage s - 1 has read
                    // (the slice base of these two or three pieces formed once per stage as well: two more live scalars, 4 VGPR spills in the planar kernel, slower)
                    if (wave < (jj == RJ - 1 ? RAW_PIECES - (RJ - 1) * DW : DW)) dma_raw(std::integral_constant<int, jj>{}, fr, r_slice);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (QT == 0 && xi < 2) {
                    raw_read(srcQ, std::integral_constant<int, 3 * xi>{});
                    raw_read(srcQ, std::integral_constant<int, 3 * xi + 1>{});
                    raw_read(srcQ, std::integral_constant<int, 3 * xi + 2>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (xi >= W4P_T0 && xi < W4P_T0 + 3) {
                    if constexpr (QT == 0 && xi == W4P_T0) raw_mask(xlimQ);
                    if constexpr (QT < 2) row_pass(std::integral_constant<int, 3 * QT + xi - W4P_T0>{});
                    else col_pass(std::integral_constant<int, 3 * (QT - 2) + xi - W4P_T0>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (QT == 3 && xi >= W4P_T0 + 3 && xi < W4P_T0 + 6) {
                    v_write(dstQ, std::integral_constant<int, 3 * (xi - (W4P_T0 + 3))>{});
                    v_write(dstQ, std::integral_constant<int, 3 * (xi - (W4P_T0 + 3)) + 1>{});
                    v_write(dstQ, std::integral_constant<int, 3 * (xi - (W4P_T0 + 3)) + 2>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        };
        load_first(0u);
        // the loop, rotated by one stage: the first stage of the item is its own copy of the stage body (five copies instead of four)
        stage(std::integral_constant<int, 0>{}, std::true_type{}, 0);
#pragma unroll 1
        for (int s = 1; s < NST; s += 4) {
            stage(std::integral_constant<int, 1>{}, std::false_type{}, s);
            stage(std::integral_constant<int, 2>{}, std::false_type{}, s + 1);
            stage(std::integral_constant<int, 3>{}, std::false_type{}, s + 2);
            if (s + 3 < NST) stage(std::integral_constant<int, 0>{}, std::false_type{}, s + 3);
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // (the last MFMAs' results: the hazard the compiler does not see through the asm)
        xlim_cur = xlim_r;   // (the raw cursor entered the next item's tile five stages ago)
        if constexpr (PH != 0) park();   // (PH = 0 has just written the V of the next item's first stage)
        {
            // ---- epilogue: Y = A^T M A, bias, LeakyReLU, stores.  C/D of the 16x16 MFMA: lane & 15 = block, register e = plane
            //      4 (lane >> 4) + e of the plane tile ----
            __builtin_amdgcn_s_setprio(2);
            const int ob = item % NOB;
            const int lane_e = lane_o(), t = lane_e & 15, k = lane_e >> 4;
            int tile_y, tile_x;
            tile_coords(item / NOB, tile_y, tile_x);
            const int ty0 = tile_y * ROWS - d.wino_py;
            const int oy = ty0 + 4 * (2 * bt + (t >> 3)), ox = tile_x * 32 + 4 * (t & 7);
            const int plane0 = ob * 64 + pt * 16 + 4 * k;
            float *obase = OUT_PLANAR ? d.out + (long long)plane0 * d.out_cs + (long long)oy * d.out_rs + ox
                                      : d.out + (long long)oy * d.out_rs + (long long)ox * COUT + plane0;
            const bool interior = ty0 >= 0 && ty0 + ROWS <= d.out_h && tile_x * 32 + 32 <= d.out_w;   // wave-uniform
            const f32x4 bq = *reinterpret_cast<const f32x4 *>(ldsb + BIAS_BASE + plane0 * 4);
            // FUSE7: the one-plane LAST layer (convertRoutine.cpp:66-76's next iteration) inside this epilogue, "taps as rows": per pixel the nine partial
            // sums T_tap = sum over this wave's 16 planes of w7[plane][tap] * a6[plane] on the MFMA (A = w7 of the planes 4 k + e as 16 x 4, tap = row;
            // B = the activations just computed, lane (k, block) = plane 4 k + e of a pixel of that block), summed over the four plane-tile waves
            // through the V slot that is idle between two items, and written as 9 tap planes per 64-plane block: 72 bytes per pixel leave the chip
            // instead of 512, and conv3x3_last_gather adds taps and blocks (0.1 ms instead of the 0.8 ms of conv3x3_last).
            float a7[4];
            if constexpr (FUSE7) {
                const float *w7 = reinterpret_cast<const float *>(d.w7pk) + (size_t)(ob * 4 + pt) * 256 + lane_e;   // [16-plane group][e][lane]
#pragma unroll
                for (int e = 0; e < 4; e++) a7[e] = w7[e * 64];
            }
            char *red = ldsb + V_BASE + V_BYTES;   // (V slot 1: idle until the next item's first stage writes V of its second)
            unsigned tick = 0xFFFFFFFFu, pq0 = 0, pq1 = 0;   // PROG, control wave: what the job counters of the PREVIOUS item's arrivals held; the queue
            (void)tick; (void)pq0; (void)pq1;
            // [pt][tap][128 pixels] floats (18 KiB)
            // The row transform A^T M of every column, all four rows at once (10 operations per column and plane; two passes over row PAIRS recompute
            // the four sums and differences: 14): the six accumulators of a (column, plane) are dead behind it, their registers hold its four results.
            float tm[4][6][4];   // [row][column j][plane e]
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    const float m0 = acc[xi_of(0, j)][e], m1 = acc[xi_of(1, j)][e], m2 = acc[xi_of(2, j)][e], m3 = acc[xi_of(3, j)][e],
                                m4 = acc[xi_of(4, j)][e], m5 = acc[xi_of(5, j)][e];
                    const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
                    tm[0][j][e] = m0 + s1 + s2;
                    tm[1][j][e] = __builtin_fmaf(1.5f, d2, 0.75f * d1);
                    tm[2][j][e] = __builtin_fmaf(2.25f, s2, 0.5625f * s1);
                    tm[3][j][e] = __builtin_fmaf(3.375f, d2, __builtin_fmaf(0.421875f, d1, m5));
                }
            // PROG: the PREVIOUS item's tap planes (and every transfer but the youngest) have long landed: this wait is all but free here, and makes
            // that item's arrival countable behind the first barrier below
            if constexpr (PROG) W2XC_WAIT_VMCNT(0);
#pragma unroll
            for (int rp = 0; rp < 2; rp++) {
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    const int i = 2 * rp + rr;
                    f32x4 y[4];      // OUT_PLANAR: y[e] = the four pixels of row i of plane e; NHWC: y[j] = the four planes of pixel j
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        float y0, y1, y2, y3;
                        at6(tm[i][0][e], tm[i][1][e], tm[i][2][e], tm[i][3][e], tm[i][4][e], tm[i][5][e], y0, y1, y2, y3);
                        const float w0 = y0 + bq[e], w1 = y1 + bq[e], w2 = y2 + bq[e], w3 = y3 + bq[e];
                        const float l0 = __builtin_amdgcn_fmed3f(w0, 0.1f * w0, 3.402823466e+38f), l1 = __builtin_amdgcn_fmed3f(w1, 0.1f * w1, 3.402823466e+38f);
                        const float l2 = __builtin_amdgcn_fmed3f(w2, 0.1f * w2, 3.402823466e+38f), l3 = __builtin_amdgcn_fmed3f(w3, 0.1f * w3, 3.402823466e+38f);
                        if constexpr (OUT_PLANAR || FUSE7) y[e] = f32x4{l0, l1, l2, l3};
                        else { y[0][e] = l0; y[1][e] = l1; y[2][e] = l2; y[3][e] = l3; }
                    }
                    if constexpr (FUSE7) {
                        // y[e] = the four pixels of row i of plane e (OUT_PLANAR form): D[j] = taps of pixel j of this lane's block over the wave's planes
                        f32x4 D[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) D[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                        for (int e = 0; e < 4; e++)
#pragma unroll
                            for (int j = 0; j < 4; j++) D[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a7[e], y[e][j], D[j], 0, 0, 0);
                        // lane (k, t) holds taps 4 k .. 4 k + 3 of pixel j of block t.  The slab is [pt][tap][128 pixels] (pixel index p = (block row) * 32 +
                        // (block column) * 4 + j): ONE 16-byte write per tap = the four pixels of this lane's block, eight lanes = 128 contiguous bytes (as
                        // [pt][tap quad][pixel][4] the eight lanes of a write group sat 64 bytes apart: four-way bank conflicts, 9 % of layer 6's LDS cycles)
                        const int p0 = (2 * bt + (t >> 3)) * 32 + (t & 7) * 4;
                        if (k < 2) {
#pragma unroll
                            for (int r = 0; r < 4; r++) *reinterpret_cast<f32x4 *>(red + ((pt * 9 + 4 * k + r) * 128 + p0) * 4) = f32x4{D[0][r], D[1][r], D[2][r], D[3][r]};
                        } else if (k == 2) {
                            *reinterpret_cast<f32x4 *>(red + ((pt * 9 + 8) * 128 + p0) * 4) = f32x4{D[0][0], D[1][0], D[2][0], D[3][0]};
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                        if constexpr (PROG) {
                            // (every wave has drained the previous item's stores in front of this barrier) its arrival, counted by wave 0; the counters'
                            // answers are looked at three row steps further down
                            if (i == 0 && wave == CW) prog_early(n > 0 ? item_of(n - 1) : -1, tick, pq0, pq1);
                            if (i == 3 && wave == CW) prog_late(n > 0 ? item_of(n - 1) : -1, tick, pq0, pq1);
                        }
                        {
                            const int tid = wave * 64 + lane_e;
                            if (tid < 288) {
                                const int tap = tid >> 5, p = (tid & 31) * 4;     // nine taps x 32 pixel quads
                                const int gy = ty0 + 4 * (p >> 5) + i, gx = tile_x * 32 + (p & 31);
                                f32x4 sum = *reinterpret_cast<const f32x4 *>(red + ((0 * 9 + tap) * 128 + p) * 4);
#pragma unroll
                                for (int q = 1; q < 4; q++) sum += *reinterpret_cast<const f32x4 *>(red + ((q * 9 + tap) * 128 + p) * 4);   // (fixed order: reproducible)
                                if (gy >= 0 && gy < d.out_h && gx < d.out_w) {
                                    float *g = d.out + (long long)ob * d.out_ts + (long long)tap * d.out_gs + (long long)gy * d.out_rs + gx;
                                    if constexpr (PROG) {   // written through: read by the gather job of whichever workgroup arrives last
                                        if (gx + 3 < d.out_w) store16_sc1(g, sum);
                                        else {
#pragma unroll
                                            for (int e = 0; e < 3; e++)
                                                if (gx + e < d.out_w) store4_sc1(g + e, sum[e]);
                                        }
                                    } else if (gx + 3 < d.out_w) *reinterpret_cast<f32x4u *>(g) = sum;   // (dword-aligned: rows of out_w floats)
                                    else {
#pragma unroll
                                        for (int e = 0; e < 3; e++)
                                            if (gx + e < d.out_w) g[e] = sum[e];
                                    }
                                }
                            }
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();   // (the slab is rewritten by the next row)
                        asm volatile("" ::: "memory");
                    } else if constexpr ((W4_ABL & 64) != 0) {
                        if (y[0][0] == 12345.678f) *reinterpret_cast<f32x4 *>(obase) = y[0] + y[1] + y[2] + y[3];
                    } else if constexpr (OUT_PLANAR) {
                        // whole quads: the row stride holds roundup4(out_w) pixels (the launcher checks), columns >= out_w are never read as data
                        if (interior || (oy + i >= 0 && oy + i < d.out_h && ox < d.out_w)) {
#pragma unroll
                            for (int e = 0; e < 4; e++) *reinterpret_cast<f32x4 *>(obase + (long long)e * d.out_cs + (long long)i * d.out_rs) = y[e];
                        }
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
            // PROG: the gather jobs the previous item's arrivals completed (the LDS word was written in front of the epilogue's last barrier)
            if constexpr (PROG) prog_run();
        }
        // the transforms in flight back into registers; the barrier: the next stage's U transfers land on the parking area
        if constexpr (PH != 0) unpark();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    W2XC_WAIT_VMCNT(0);   // drain the speculative transfers before the LDS is released
    if constexpr (PROG) {
        // the workgroup's LAST item: its tap planes have drained (the wait above, every wave), its arrival is counted and what it completed is pushed; then
        // the workgroup works the queue off -- this once with every latency exposed -- until it holds no ticket and sees no ready job without one.  Jobs pushed
        // later are drawn by workgroups still running or by their pusher's own pass through here; a ticket whose slot is still empty is waited for (the slot
        // is filled by a workgroup that is running and waits for nobody).
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        bool first = true;
        for (;;) {
            if (wave == CW) {
                unsigned tick, q0, q1;
                prog_early(first ? item_of(nmy - 1) : -1, tick, q0, q1);
                prog_late(first ? item_of(nmy - 1) : -1, tick, q0, q1);
                unsigned t = *lds_ticket;
                if (*lds_action == 0 && t) {   // a ticket in hand: wait for its slot
                    unsigned q = 0;
                    while ((q = __hip_atomic_load(q_slots + (t - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) __builtin_amdgcn_s_sleep(8);
                    if (lane_o() == 0) { *lds_action = q; *lds_ticket = 0; }
                }
            }
            const bool was_first = first;
            first = false;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const unsigned ran = prog_run();
            __builtin_amdgcn_s_barrier();   // (the control wave rewrites the words)
            asm volatile("" ::: "memory");
            // (the first pass looked at the queue BEFORE it pushed what this workgroup's last arrival completed: the pass that ends the loop is one that
            //  has seen the queue after every push of this workgroup -- or the last pusher of the launch would leave with its job still in the queue)
            if (!ran && !was_first) break;
        }
    }
    };
#ifdef W4_ONE_PH   // timing-only (tools/ubench): every wave runs the copy of phase W4_ONE_PH -- wrong results, the same work per stage, a quarter of the hot code
    run(std::integral_constant<int, W4_ONE_PH>{});
    return;
#endif
    switch ((pt - 2 * bt) & 3) {
    case 0: run(std::integral_constant<int, 0>{}); break;
    case 1: run(std::integral_constant<int, 1>{}); break;
    case 2: run(std::integral_constant<int, 2>{}); break;
    default: run(std::integral_constant<int, 3>{}); break;
    }
}

// ------------------------------------------------------------------------------------------------
// host side.  Three objects (make -j): W2XC_WINO4_PART = 0 the planar-out instantiations + packers + dispatcher, 1 the NHWC-out ones, 2 the fused-last ones.
// ------------------------------------------------------------------------------------------------
#ifndef W2XC_WINO4_PART
#define W2XC_WINO4_PART -1   // one translation unit with everything (tools/ubench)
#endif
template <int CIN, int COUT, bool OUT_PLANAR, bool IN_NHWC = false, bool FUSE7 = false, bool PROG = false>
static hipError_t launch_wino4(const W2xcConvDesc &d, hipStream_t stream)
{
    const int tiles_x = (d.out_w + 31) / 32, tiles_y = (d.out_h + (d.wino_py & 3) + 15) / 16;
    const int nitems = tiles_x * tiles_y * (COUT / 64);
    constexpr size_t lds_bytes = 3 * (size_t)(11 * 1024) + 2 * (size_t)(36 * 1024) + 3 * (size_t)(18 * 1024) + COUT * 4 + (PROG ? 16 : 0);   // raw + U + V + bias (+ the job word)
    static_assert(lds_bytes <= 160 * 1024, "LDS budget");
    if constexpr (PROG) {
        // the job counters start at zero in EVERY launch (a memset node in front of the kernel, on its stream)
        if (!d.prog_cnt || !d.g_out || !d.g_bias || d.g_w < 4 || (long long)d.out_h * d.out_rs * 4 >= (1ll << 32)) return hipErrorInvalidValue;   // (32-bit offsets inside a tap plane, whole quads)
        hipError_t em = hipMemsetAsync(d.prog_cnt, 0, w2xc_wino4_prog_counters(d.out_w, d.out_h, d.wino_py) * sizeof(unsigned), stream);
        if (em != hipSuccess) return em;
    }
    auto kern = conv3x3_wino4<CIN, COUT, OUT_PLANAR, IN_NHWC, FUSE7, PROG>;
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


hipError_t w2xc_launch_wino4_nhwc_out(const W2xcConvDesc &d, hipStream_t stream);
#if W2XC_WINO4_PART != 0 && W2XC_WINO4_PART != 2
hipError_t w2xc_launch_wino4_nhwc_out(const W2xcConvDesc &d, hipStream_t stream)
{
#ifdef W4P_SINGLE
    return hipErrorInvalidValue;
#else
    if (d.cin == 32 && d.in_ps == 32 && d.in_cs == 1)
        return d.cout == 64 ? launch_wino4<32, 64, false, true>(d, stream) : d.cout == 128 ? launch_wino4<32, 128, false, true>(d, stream) : hipErrorInvalidValue;
    switch (d.cin * 1000 + d.cout) {
    case 32064:  return launch_wino4<32, 64, false>(d, stream);
    case 32128:  return launch_wino4<32, 128, false>(d, stream);
    case 64064:  return launch_wino4<64, 64, false>(d, stream);
    case 64128:  return launch_wino4<64, 128, false>(d, stream);
    case 128064: return launch_wino4<128, 64, false>(d, stream);
    case 128128: return launch_wino4<128, 128, false>(d, stream);
    default: return hipErrorInvalidValue;
    }
#endif
}
#endif

// d.out_terms = 9: the one-plane last layer in the epilogue; `out` = partial tap planes G[64-plane block][tap][y][x] (out_ts / out_gs / out_rs),
// d.w7pk = the w2xc_wino4_pack_last image of its weights; W2XC_K_LAST_GATHER finishes with halves = cout / 64
hipError_t w2xc_launch_wino4_fused(const W2xcConvDesc &d, hipStream_t stream);
#if W2XC_WINO4_PART != 0 && W2XC_WINO4_PART != 1
hipError_t w2xc_launch_wino4_fused(const W2xcConvDesc &d, hipStream_t stream)
{
#ifdef W4P_SINGLE
    if (d.prog_cnt) return d.cin == 128 && d.cout == 128 && d.in_ps == 1 ? launch_wino4<128, 128, true, false, true, true>(d, stream) : hipErrorInvalidValue;
    return d.cin == 128 && d.cout == 128 && d.in_ps == 1 ? launch_wino4<128, 128, true, false, true>(d, stream) : hipErrorInvalidValue;
#else
    if (d.prog_cnt) {   // the launch finishes the last layer itself (PROG); planar 64 / 128-plane inputs (the 7-layer models' layer n - 1)
        if (d.in_ps != 1) return hipErrorInvalidValue;
        switch (d.cin * 1000 + d.cout) {
        case 64064:  return launch_wino4<64, 64, true, false, true, true>(d, stream);
        case 64128:  return launch_wino4<64, 128, true, false, true, true>(d, stream);
        case 128064: return launch_wino4<128, 64, true, false, true, true>(d, stream);
        case 128128: return launch_wino4<128, 128, true, false, true, true>(d, stream);
        default: return hipErrorInvalidValue;
        }
    }
    if (d.cin == 32 && d.in_ps == 32 && d.in_cs == 1)
        return d.cout == 64 ? launch_wino4<32, 64, true, true, true>(d, stream) : d.cout == 128 ? launch_wino4<32, 128, true, true, true>(d, stream) : hipErrorInvalidValue;
    switch (d.cin * 1000 + d.cout) {
    case 64064:  return launch_wino4<64, 64, true, false, true>(d, stream);
    case 64128:  return launch_wino4<64, 128, true, false, true>(d, stream);
    case 128064: return launch_wino4<128, 64, true, false, true>(d, stream);
    case 128128: return launch_wino4<128, 128, true, false, true>(d, stream);
    default: return hipErrorInvalidValue;
    }
#endif
}
#endif

#if W2XC_WINO4_PART == 0 || W2XC_WINO4_PART == -1
// the last layer's weights as MFMA A fragments for conv3x3_wino4's fused epilogue: [16-plane group g][e][lane = 16 kk + m] = w7[plane 16 g + 4 kk + e][tap m]
// (m < 9, else 0).  w is [1][cin][3][3] (modelHandler.cpp:102).  16 * cin floats.
size_t w2xc_wino4_pack_last_floats(int cin) { return (size_t)16 * cin; }
void w2xc_wino4_pack_last(int cin, const float *w, float *dst)
{
    for (int g = 0; g < cin / 16; g++)
        for (int e = 0; e < 4; e++)
            for (int kk = 0; kk < 4; kk++)
                for (int m = 0; m < 16; m++) dst[((size_t)g * 4 + e) * 64 + kk * 16 + m] = m < 9 ? w[(size_t)(16 * g + 4 * kk + e) * 9 + m] : 0.0f;
}
#endif

#if W2XC_WINO4_PART != 1 && W2XC_WINO4_PART != 2
bool w2xc_wino4_supported(int cin, int cout)
{
    return (cin == 32 || cin == 64 || cin == 128) && (cout == 64 || cout == 128);
}
// PROG: the fused launch that finishes the last layer itself exists for planar 64 / 128-plane inputs
bool w2xc_wino4_prog_supported(int cin, int cout) { return (cin == 64 || cin == 128) && (cout == 64 || cout == 128); }
// ... and its control words: per job (tile row, group of 8 tile columns of the launch's region) an arrival counter and a queue slot, + head and tail
void w2xc_wino4_prog_jobs(int out_w, int out_h, int wino_py, int *tile_rows, int *groups)
{
    *tile_rows = (out_h + (wino_py & 3) + 15) / 16;
    *groups = ((out_w + 31) / 32 + 7) / 8;
}
size_t w2xc_wino4_prog_counters(int out_w, int out_h, int wino_py)
{
    const int tiles_x = (out_w + 31) / 32, tiles_y = (out_h + (wino_py & 3) + 15) / 16;
    return 2 * (size_t)tiles_y * ((tiles_x + 7) / 8) + 2;   // arrivals per job | the ready queue | head, tail
}

// wpk[64-plane block ob][stage s (4 channels)][xi / 4][plane tile pt][lane = 16 k + o][xi % 4] = U_xi[plane 64 ob + 16 pt + o][channel 4 s + k], xi = xi_of(i, j),
// U = G g G^T formed in double and rounded once.  w is [cout][cin][3][3] (modelHandler.cpp:102).  36 * cin * cout floats.
void w2xc_wino4_pack(int cin, int cout, const float *w, float *dst)
{
    static const double GM[6][3] = {{64.0 / 81, 0, 0},
                                    {-128.0 / 243, -32.0 / 81, -8.0 / 27},
                                    {-128.0 / 243, 32.0 / 81, -8.0 / 27},
                                    {32.0 / 243, 16.0 / 81, 8.0 / 27},
                                    {32.0 / 243, -16.0 / 81, 8.0 / 27},
                                    {0, 0, 1}};
    const int nst = cin / 4, nob = cout / 64;
    for (int ob = 0; ob < nob; ob++)
        for (int s = 0; s < nst; s++)
            for (int pt = 0; pt < 4; pt++)
                for (int k = 0; k < 4; k++)
                    for (int o = 0; o < 16; o++) {
                        const int plane = 64 * ob + 16 * pt + o, c = 4 * s + k;
                        const float *g = w + ((size_t)plane * cin + c) * 9;
                        double tmp[6][3];
                        for (int i = 0; i < 6; i++)
                            for (int j = 0; j < 3; j++) tmp[i][j] = GM[i][0] * g[0 * 3 + j] + GM[i][1] * g[1 * 3 + j] + GM[i][2] * g[2 * 3 + j];
                        for (int i = 0; i < 6; i++)
                            for (int j = 0; j < 6; j++) {
                                const double u = tmp[i][0] * GM[j][0] + tmp[i][1] * GM[j][1] + tmp[i][2] * GM[j][2];
                                const int xi = xi_of(i, j);
                                dst[(((((size_t)ob * nst + s) * 9 + (xi >> 2)) * 4 + pt) * 64 + k * 16 + o) * 4 + (xi & 3)] = (float)u;
                            }
                    }
}


// d.wpk = w2xc_wino4_pack image; planar fp32 in (in_ps = 1, in_cs = plane stride), planar (out_ps = 1) or NHWC (out_cs = 1, out_ps = cout) out;
// d.wino_py = first output row mod 4; off_x a multiple of 4 (the engine's layers: 0)
hipError_t w2xc_launch_wino4(const W2xcConvDesc &d, hipStream_t stream)
{
    if (d.out_w <= 0 || d.out_h <= 0) return hipSuccess;
    const bool in_nhwc = d.cin == 32 && d.in_ps == 32 && d.in_cs == 1;   // the 32-plane layers in front write NHWC
    if (d.in_shift != 0 || (d.in_rs & 3) != 0 || (((size_t)d.in) & 15) != 0) return hipErrorInvalidValue;
    if (in_nhwc) {
        if (24 * d.in_rs * 4 >= (1ll << 32)) return hipErrorInvalidValue;
    } else {
        if (d.in_ps != 1 || (d.in_cs & 3) != 0 || d.off_x < 0 || (d.off_x & 3) != 0 || d.in_rs < ((d.in_w + 3) & ~3)) return hipErrorInvalidValue;
        if (3 * d.in_cs * 4 + 24 * d.in_rs * 4 >= (1ll << 32)) return hipErrorInvalidValue;   // 32-bit lane offsets inside a 4-channel slice of a tile
    }
    if (d.out_terms == 9) {   // fused last layer: partial tap planes, rows of out_w floats
        if (d.cout != 64 && d.cout != 128) return hipErrorInvalidValue;
        if (!d.w7pk || d.out_rs < d.out_w || d.out_gs < d.out_rs * (long long)d.out_h) return hipErrorInvalidValue;
        return w2xc_launch_wino4_fused(d, stream);
    }
    const bool planar = d.out_ps == 1;
    if (planar) {
        if ((d.out_rs & 3) != 0 || (d.out_cs & 3) != 0 || (((size_t)d.out) & 15) != 0 || d.out_rs < ((d.out_w + 3) & ~3)) return hipErrorInvalidValue;
    } else if (d.out_ps != d.cout || d.out_cs != 1 || (d.out_rs & 3) != 0 || (((size_t)d.out) & 15) != 0) return hipErrorInvalidValue;
    if (!planar) return w2xc_launch_wino4_nhwc_out(d, stream);
#ifndef W4P_SINGLE
    if (in_nhwc) return d.cout == 64 ? launch_wino4<32, 64, true, true>(d, stream) : d.cout == 128 ? launch_wino4<32, 128, true, true>(d, stream) : hipErrorInvalidValue;
#endif
#ifdef W4P_SINGLE   // (development builds: one instantiation)
    return d.cin == 128 && d.cout == 128 ? launch_wino4<128, 128, true>(d, stream) : hipErrorInvalidValue;
#else
    switch (d.cin * 1000 + d.cout) {
    case 32064:  return launch_wino4<32, 64, true>(d, stream);
    case 32128:  return launch_wino4<32, 128, true>(d, stream);
    case 64064:  return launch_wino4<64, 64, true>(d, stream);
    case 64128:  return launch_wino4<64, 128, true>(d, stream);
    case 128064: return launch_wino4<128, 64, true>(d, stream);
    case 128128: return launch_wino4<128, 128, true>(d, stream);
    default: return hipErrorInvalidValue;
    }
#endif
}
#endif
