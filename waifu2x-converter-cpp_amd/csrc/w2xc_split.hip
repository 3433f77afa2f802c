// w2xc_split.hip -- "split" kernels for gfx950: the 3x3xCinxCout contraction of Model::filterWorker
// (/root/reference/src/modelHandler.cpp:117-159) on the 16-bit MFMAs (v_mfma_f32_32x32x16_bf16 / _f16: 2.5 PFLOP/s
// dense, 16x the fp32 MFMA rate of CDNA4) with every fp32 operand carried as a sum of T 16-bit terms:
//
//     a = a0 + a1 (+ a2),  a0 = rnd16(a), a1 = rnd16(a - a0), a2 = rnd16(a - a0 - a1)      (round to nearest even)
//     a*b ~= sum of the NP largest term products, accumulated in fp32 inside the matrix core
//
//     T = 2 bf16  (W2XC_PRECISION_BF16X2)  3 products  a1b0 + a0b0 + a0b1            ~16-bit operands, |err| ~ 2^-17 |ab|
//     T = 3 bf16  (W2XC_PRECISION_BF16X3)  6 products  + a1b1 + a2b0 + a0b2          ~24-bit operands, |err| ~ 2^-24 |ab|,
//                                                                                    the error level of an fp32 FMA chain
//     T = 2 fp16  (W2XC_PRECISION_FP16X2, FMT = 1)  3 products                       ~22-bit operands; weights pre-scaled per
//                 layer by a power of two (w2xc_split_pack), activations clamped to +-65504
//     (T = 1 compiles but is not instantiated: W2XC_PRECISION_BF16 keeps its own kernels in w2xc_kernels.hip.)
//
// Activations between the layers are T "term planes" (`ts` elements apart), each channel-group blocked:
// element (c, y, x) at (c / 16)*gs + y*rs + x*16 + c % 16, so the halo tile of one (16-channel slice, term) is
// contiguous per row.  The producer's epilogue does the split once per element, so the consumer streams ready-made
// 16-bit fragments with LDS-DMA exactly like conv3x3_mfma2 streams fp32 ones.
//
//   conv3x3_split        cin, cout in {32,64,128}: persistent workgroups (one per CU), tile = 8 or 16 rows x 32 pixels
//                        x COUT.  Stage = (16-channel slice, tap); LDS = A[2] (halo tile of one slice, all T terms, 32
//                        bytes per pixel per term, the two 16-byte chunks XOR-swizzled so ds_read_b128 of 32 consecutive
//                        pixels is conflict-free) + a ring of RING B stages (T*COUT/32 KiB each, fragment order) + bias.
//                        Operands are swapped (weights = MFMA A operand): the accumulator tile is [channel][pixel],
//                        lane = pixel, 4 consecutive channels per register quad -> 8/16-byte stores, no LDS transpose.
//                        OT = T: term planes out; OT = 0: fp32 NHWC out (feeds conv3x3_last); OT = 9 (two terms): the
//                        one-plane LAST layer is computed in the epilogue from the accumulator registers and its partial
//                        tap planes go to conv3x3_last_gather.
//   conv3x3_first_split  cin <= 3 (layer 1) on the fp32 MFMA exactly like conv3x3_first, storing term planes.
#include "w2xc_kernels.h"
#include "w2xc_device.h"

#include <stdlib.h>
#include <math.h>
#include <string.h>

#include <atomic>
#include <type_traits>

#ifndef W2XC_SPLIT_T
#error "compile with -DW2XC_SPLIT_T=1, 2, 3, 4 (= fp16 x 2) or 5 (= 3 terms, fp32 / fused-last out): one object per variant, see the Makefile"
#endif
#ifndef W2XC_SPLIT_LATE
#define W2XC_SPLIT_LATE 4   // MFMAs kept after the last fragment read of a step
#endif

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// v_cvt_pk_bf16_f32 (round to nearest even): a -> bits 0..15, b -> bits 16..31
static __device__ __forceinline__ unsigned pk_bf16(float a, float b)
{
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

typedef _Float16 h16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
// v_cvt_pk_f16_f32 (round to nearest even)
static __device__ __forceinline__ unsigned pk_f16(float a, float b)
{
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h16x2_t));
}

// LeakyReLU(0.1) of an MFMA accumulator register: median(s, 0.1 s, FLT_MAX) = max(s, 0.1 s) for every finite s.
// (fmaxf on a raw accumulator costs a second v_max_f32 -- the IEEE-mode canonicalisation of an operand the compiler cannot
// prove quiet; v_med3_f32 has no such requirement.)
static __device__ __forceinline__ float leaky_acc(float s) { return __builtin_amdgcn_fmed3f(s, 0.1f * s, 3.402823466e+38f); }   // (FLT_MAX: with +inf the compiler folds it back to fmaxf)

// v - (float)h for an fp16 h: one v_fma_mix_f32 (the conversion rides in the instruction) instead of v_cvt_f32_f16 + v_sub_f32; exact either way
// (HI = 0 / 1: the low / high half of the packed pair p.  The compiler folds fma(fpext(h), -1, v) back into the two-instruction form.)
template <int HI> static __device__ __forceinline__ float sub_f16(float v, unsigned p)
{
    float r;
    if (HI) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(v));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(v));
    return r;
}

// Store 4 consecutive channels of one pixel as OT term planes (OT >= 1) or as fp32 (OT == 0).
// FMT = 0: bf16 terms (same exponent range as fp32).  FMT = 1: fp16 terms -- values are clamped to the fp16
// range (+-65504) first; residuals below 2^-14 are held to 2^-25 absolute by fp16's subnormals.
typedef __attribute__((address_space(1))) char gbyte_t;   // a byte of global memory
// `base` = byte address of the quad's first element in term plane 0 (fp32 NHWC when OT == 0), `ts_bytes` = bytes between term planes.
template <int OT, int FMT>
static __device__ __forceinline__ void store_terms_at(gbyte_t *base, long long ts_bytes, float v0, float v1, float v2, float v3)
{
    if (OT == 0 || OT == 9) {   // (OT == 9, the fused-last-layer epilogue, never gets here)
        *(__attribute__((address_space(1))) f32x4 *)(base) = (f32x4){v0, v1, v2, v3};
    } else {
        if (FMT == 1) {
            v0 = __builtin_amdgcn_fmed3f(v0, -65504.0f, 65504.0f);
            v1 = __builtin_amdgcn_fmed3f(v1, -65504.0f, 65504.0f);
            v2 = __builtin_amdgcn_fmed3f(v2, -65504.0f, 65504.0f);
            v3 = __builtin_amdgcn_fmed3f(v3, -65504.0f, 65504.0f);
        }
#pragma unroll
        for (int t = 0; t < OT; t++) {
            const unsigned p01 = FMT ? pk_f16(v0, v1) : pk_bf16(v0, v1), p23 = FMT ? pk_f16(v2, v3) : pk_bf16(v2, v3);
            *(__attribute__((address_space(1))) u32x2 *)(base + (long long)t * ts_bytes) = (u32x2){p01, p23};
            if (t + 1 < OT) {   // exact residuals: |v - round(v)| fits fp32
                if (FMT) {
                    v0 = sub_f16<0>(v0, p01); v1 = sub_f16<1>(v1, p01); v2 = sub_f16<0>(v2, p23); v3 = sub_f16<1>(v3, p23);
                } else {
                    v0 -= __uint_as_float(p01 << 16);
                    v1 -= __uint_as_float(p01 & 0xFFFF0000u);
                    v2 -= __uint_as_float(p23 << 16);
                    v3 -= __uint_as_float(p23 & 0xFFFF0000u);
                }
            }
        }
    }
}
template <int OT, int FMT>
static __device__ __forceinline__ void store_terms(float *out, long long elem_off, long long out_ts, float v0, float v1, float v2, float v3)
{
    if (OT == 0 || OT == 9) {   // (OT == 9, the fused-last-layer epilogue, never gets here)
        *reinterpret_cast<f32x4 *>(out + elem_off) = (f32x4){v0, v1, v2, v3};
    } else {
        bf16_t *o16 = reinterpret_cast<bf16_t *>(out) + elem_off;
        if (FMT == 1) {
            v0 = __builtin_amdgcn_fmed3f(v0, -65504.0f, 65504.0f);
            v1 = __builtin_amdgcn_fmed3f(v1, -65504.0f, 65504.0f);
            v2 = __builtin_amdgcn_fmed3f(v2, -65504.0f, 65504.0f);
            v3 = __builtin_amdgcn_fmed3f(v3, -65504.0f, 65504.0f);
        }
#pragma unroll
        for (int t = 0; t < OT; t++) {
            const unsigned p01 = FMT ? pk_f16(v0, v1) : pk_bf16(v0, v1), p23 = FMT ? pk_f16(v2, v3) : pk_bf16(v2, v3);
            *reinterpret_cast<u32x2 *>(o16 + (long long)t * out_ts) = (u32x2){p01, p23};
            if (t + 1 < OT) {   // exact residuals: |v - round(v)| fits fp32
                if (FMT) {
                    v0 = sub_f16<0>(v0, p01); v1 = sub_f16<1>(v1, p01); v2 = sub_f16<0>(v2, p23); v3 = sub_f16<1>(v3, p23);
                } else {
                    v0 -= __uint_as_float(p01 << 16);
                    v1 -= __uint_as_float(p01 & 0xFFFF0000u);
                    v2 -= __uint_as_float(p23 << 16);
                    v3 -= __uint_as_float(p23 & 0xFFFF0000u);
                }
            }
        }
    }
}

// term products in issue order: activation term a(i) x weight term b(i)
template <int T> struct Prod;
template <> struct Prod<1> {
    static constexpr int N = 1;
    static __device__ constexpr int a(int) { return 0; }
    static __device__ constexpr int b(int) { return 0; }
};
template <> struct Prod<2> {   // (x1,w0), (x0,w0), (x0,w1): each fragment group is live in consecutive products (rolling registers)
    static constexpr int N = 3;
    static __device__ constexpr int a(int i) { return i == 0 ? 1 : 0; }
    static __device__ constexpr int b(int i) { return i == 2 ? 1 : 0; }
};
template <> struct Prod<3> {
    static constexpr int N = 6;
    //                                          i:   0  1  2  3  4  5
    static __device__ constexpr int a(int i) { return i == 0 ? 1 : i == 1 ? 2 : i == 3 ? 1 : 0; }   // 1  2  0  1  0  0
    static __device__ constexpr int b(int i) { return i == 0 ? 1 : i == 2 ? 2 : i == 4 ? 1 : 0; }   // 1  0  2  0  1  0
};

// Fragment read order of a step = order of first use by the products above, so the fragments read last
// are needed last: group gi of 2T is (weights? , term).  read_decode(r) -> isW*100 + term*10 + index.
template <int T> static __device__ constexpr int grp_w(int gi) { return gi & 1; }
template <int T> static __device__ constexpr int grp_term(int gi)
{
    return T == 3 ? (gi == 0 ? 1 : gi == 1 ? 1 : gi == 2 ? 2 : gi == 3 ? 0 : gi == 4 ? 0 : 2)
         : T == 2 ? (gi == 0 ? 1 : gi == 1 ? 0 : gi == 2 ? 0 : 1)
                  : 0;
}
template <int T, int MB, int NB> static __device__ constexpr int read_decode(int r)
{
    for (int gi = 0; gi < 2 * T; gi++) {
        const int n = grp_w<T>(gi) ? NB : MB;
        if (r < n) return grp_w<T>(gi) * 100 + grp_term<T>(gi) * 10 + r;
        r -= n;
    }
    return -1;
}

// A-piece schedule: APW pieces spread (ceil first) over taps 0..LASTA
template <int APW, int LASTA> static __device__ constexpr int ka(int t) { return t <= LASTA ? (APW + LASTA - t) / (LASTA + 1) : 0; }
template <int APW, int LASTA> static __device__ constexpr int ka_before(int tap)
{
    int s = 0;
    for (int t = 0; t < tap; t++) s += ka<APW, LASTA>(t);
    return s;
}
template <int APW, int LASTA> static __device__ constexpr int ka_window(int tap, int look)   // taps tap-look+1 .. tap (mod 9)
{
    int s = 0;
    for (int k = 0; k < look; k++) s += ka<APW, LASTA>((tap - k + 9) % 9);
    return s;
}

// ------------------------------------------------------------------------------------------------
// E = stages per EPOCH: the workgroup barrier (and the counted vmcnt wait in front of it) closes every E-th stage instead of
// every stage.  A 16-bit stage is 16-48 MFMAs of 32 cycles per wave -- 0.5 to 1.5k cycles, an eighth of an fp32 stage -- so a
// barrier per stage costs these kernels what it never cost conv3x3_mfma2.  Inside an epoch the waves drift up to E stages
// apart, which the ring has to absorb:  B(t + LEAD) is issued during stage t with LEAD = RING - E (the slot it overwrites
// belongs to a stage of an EARLIER epoch, which every wave has left), and the barrier that closes stage t needs B(t + E + 1)
// landed (the last step of a stage already reads the next stage's first fragments), i.e. the transfer issued
// LOOK = RING - 2E - 1 stages ago.  E = 1 is the original schedule; E = 3 (9 taps = 3 epochs per slice) wants RING = 8.
template <int CIN, int COUT, int MB, int NB, int WM, int WN, int T, int OT, int KG, int RING, int FMT, int E = 1>
__global__ void __launch_bounds__(WM *WN * 64, WM *WN / 4) conv3x3_split(W2xcConvDesc d, int tiles_x, int ntiles)
{
    constexpr int ROWS = MB * WM;                     // 8 or 16 output rows per tile
    constexpr int HW = 34, HH = ROWS + 2, NPIX = HH * HW, NPIXP = (NPIX + 31) / 32 * 32;   // 340 -> 352, 612 -> 640
    constexpr int NCH = 2 * KG;                       // 16-byte chunks per pixel per term in one slice
    constexpr int PXB = 32 * KG;                      // bytes per pixel per term
    constexpr int SWS = NCH == 2 ? 3 : NCH == 4 ? 2 : 1;   // swizzle: chunk q of pixel p sits at q ^ ((p >> SWS) & (NCH-1))
    constexpr int GRP = 16;                           // channel-group size of the blocked term-plane layout
    constexpr int SLC = GRP * KG;                     // channels per slice (KG groups: chunk q of a pixel lives in group q >> 1)
    constexpr int NSL = CIN / SLC, NBT = COUT / 32;
    constexpr int NW = WM * WN;
    constexpr unsigned A_TERM = NPIXP * PXB;          // bytes of one term of the halo tile (pixels padded to whole 1 KiB pieces)
    constexpr int A_SLOTS = T * NPIXP * NCH;          // 16-byte slots
    constexpr int A_PIECES = A_SLOTS / 64;            // 1 KiB pieces = T * 11 * KG
    constexpr int APW = (A_PIECES + NW - 1) / NW;     // pieces per wave per slice
    constexpr unsigned A_BYTES = NW * APW * 1024;
    constexpr int B_PIECES = T * KG * NBT;            // 1 KiB pieces per stage
    constexpr int BPW = (B_PIECES + NW - 1) / NW;
    constexpr unsigned B_BYTES = B_PIECES * 1024;
    constexpr unsigned B_BASE = 2 * A_BYTES;
    constexpr int NP = Prod<T>::N;
    constexpr int LEAD = RING - E;                    // B(t + LEAD) is issued during stage t
    constexpr int LOOK = RING - 2 * E - 1;            // stages whose transfers are younger than B(t+E+1) at the barrier that closes stage t
    constexpr int LASTA = 9 - LEAD < 5 ? 9 - LEAD : 5;     // A pieces of the next slice are issued on taps 0..LASTA: landed by the last barrier
                                                           // before tap 8 ends (tap 8's last step already reads the next slice's first
                                                           // fragments): that barrier waits for B(9), issued in tap 9 - LEAD after the tap's A pieces
    constexpr int NST = OT == 9 ? MB * 5 : MB * NB * 4 * (OT ? OT : 1);  // store instructions of an interior-tile epilogue
    static_assert((NW == 4 || NW == 8) && (ROWS == 8 || ROWS == 16) && NB * WN == NBT, "tile shape");
    static_assert(CIN % (16 * KG) == 0 && COUT % 32 == 0 && A_SLOTS % 64 == 0, "planes");
    static_assert((E == 1 || E == 3) && RING >= 2 * E + 2 && RING <= 12 && LEAD <= 9 && (APW + LASTA) / (LASTA + 1) <= 4, "pipeline shape");
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

    // bias[COUT] sits behind the weight ring; lane (li, kk) reads the 4 channels of a register quad as one b128
    // 8-wave shapes (accumulators in VGPRs, two waves per SIMD): the accumulators START at the bias -- a ds_read_b128 per register
    // quad straight into the accumulator registers replaces a v_mov to zero them plus a v_add in the epilogue, 2 of the 4.5 VALU
    // instructions per element of an epilogue that both waves of a SIMD reach together.  fp16 weights are pre-scaled by
    // S = 1 / acc_scale (a power of two): the bias is too.  The 4-wave shapes keep their accumulators in AGPRs (zeroed for free by the
    // first MFMA of a tile, and a load would need a v_accvgpr_write per register): they add the bias in the epilogue as before.
    constexpr bool BINIT = (NW == 8);
    constexpr unsigned BIAS_BASE = B_BASE + RING * B_BYTES;
    {
        const float bmul = (BINIT && FMT) ? 1.0f / d.acc_scale : 1.0f;
        for (int c = threadIdx.x; c < COUT; c += NW * 64) lds[BIAS_BASE / 4 + c] = d.bias[c] * bmul;   // visible after the prologue barrier
    }
    // OT == 9 (last layer fused into this epilogue): its weights as MFMA A fragments, [term][plane block][k-group][lane][8]
    // LT = terms of the fused product: 1 (one product) in the one-term mode, 2 (3 products) in the two-term modes, 3 (6) for BF16X3.
    constexpr int LT = T == 3 ? 3 : T == 1 ? 1 : 2;
    constexpr unsigned W7_BASE = BIAS_BASE + COUT * 4;
    constexpr bool W7_IN_LDS = (T == 2 && E == 1) || T == 1;   // (the three-term and two-term 8-slot shapes have no LDS left: fragments come from L2)
    if constexpr (OT == 9 && W7_IN_LDS) {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(d.w7pk);
        u32x4 *dst = reinterpret_cast<u32x4 *>(const_cast<char *>(ldsb) + W7_BASE);
        for (int i = threadIdx.x; i < LT * NBT * 2 * 64; i += NW * 64) dst[i] = src[i];
    }

    // ---- per-lane DMA source offsets of the A halo tile, in 16-byte units (8 bf16) ----
    const u32x4 *in4 = reinterpret_cast<const u32x4 *>(d.in);
    const unsigned ts16 = (unsigned)(d.in_ts >> 3), gs16 = (unsigned)(d.in_gs >> 3);   // term / channel-group strides
    unsigned goff[APW], lofs[APW];
    auto slot_of = [&](int jj, int &t, int &p, int &q) {
        int s = (jj * NW + wave) * 64 + lane;
        s = s < A_SLOTS ? s : A_SLOTS - 1;                   // pieces past the data re-read its last slot
        t = s / (NPIXP * NCH);
        const int rem = s - t * (NPIXP * NCH);
        p = rem / NCH;
        const int qq = rem - p * NCH;
        p = p < NPIX ? p : NPIX - 1;                         // pad pixels re-read the last pixel
        q = qq ^ ((p >> SWS) & (NCH - 1));                   // chunk stored at this position
    };
#pragma unroll
    for (int jj = 0; jj < APW; jj++) {
        int t, p, q;
        slot_of(jj, t, p, q);
        const int py = p / HW, px = p - py * HW;
        lofs[jj] = (unsigned)t * ts16 + (unsigned)(q >> 1) * gs16 + (unsigned)(((long long)py * d.in_rs + (long long)px * GRP) >> 3) + (q & 1);
    }
    auto tile_offsets = [&](int tl) {
        const int ty_ = tl / tiles_x, tx_ = tl - ty_ * tiles_x;
        const int y0 = ty_ * ROWS + d.off_y, x0 = tx_ * 32 + d.off_x;
        if (y0 >= 0 && y0 + HH <= d.in_h && x0 >= 0 && x0 + HW <= d.in_w) {   // wave-uniform
            const unsigned base = (unsigned)(((long long)y0 * d.in_rs + (long long)x0 * GRP) >> 3);
#pragma unroll
            for (int jj = 0; jj < APW; jj++) goff[jj] = lofs[jj] + base;
            return;
        }
#pragma unroll
        for (int jj = 0; jj < APW; jj++) {
            int t, p, q;
            slot_of(jj, t, p, q);
            const int py = p / HW, px = p - py * HW;
            const int gy = clampi(y0 + py, 0, d.in_h - 1);
            const int gx = clampi(x0 + px, 0, d.in_w - 1);
            goff[jj] = (unsigned)t * ts16 + (unsigned)(q >> 1) * gs16 + (unsigned)(((long long)gy * d.in_rs + (long long)gx * GRP) >> 3) + (q & 1);
        }
    };
    auto dma_a = [&](unsigned add, unsigned abuf, int jj) {
        lds_dma16(in4 + goff[jj] + add, lds0 + abuf * A_BYTES + (unsigned)(jj * NW + wave) * 1024u);
    };
    // B piece jb of this wave for stage (sl_, tap_) -> ring slot buf (a short last wave repeats the last piece)
    const unsigned b_voff = (unsigned)lane * 16u;
    auto dma_b = [&](int sl_, int tap_, unsigned buf, int jb) {
        int pb = wave * BPW + jb;
        pb = pb < B_PIECES ? pb : B_PIECES - 1;
        // (wave-uniform by construction; readfirstlane keeps the base and the LDS address in SGPRs)
        const unsigned goff_b = (unsigned)__builtin_amdgcn_readfirstlane(((tap_ * NSL + sl_) * B_PIECES + pb) * 1024);
        const char *sbase = reinterpret_cast<const char *>(d.wpk) + goff_b;
        lds_dma16_s<0>(sbase, b_voff, (unsigned)__builtin_amdgcn_readfirstlane(lds0 + B_BASE + buf * B_BYTES + (unsigned)pb * 1024u));
    };

    // ---- fragment addressing ----
    // X (pixels): lane (li, kk) reads chunk 2g + kk of pixel p = (wm*MB + row)*34 + li + tx, term t:
    //    byte  t*A_TERM + p*PXB + (((2g + kk) ^ sw(p)) << 4)  =  t*A_TERM + (a0[row][tx] ^ (g << 5))
    unsigned a0[MB + 2][3];
#pragma unroll
    for (int row = 0; row < MB + 2; row++)
#pragma unroll
        for (int tx = 0; tx < 3; tx++) {
            const int p = (wm * MB + row) * HW + li + tx;
            a0[row][tx] = (unsigned)(p * PXB + ((((p >> SWS) & (NCH - 1)) ^ kk) << 4));
        }
    auto x_addr = [&](unsigned abuf, int t, int mb, int tap, int g) -> const u32x4 * {
        return reinterpret_cast<const u32x4 *>(ldsb + ((a0[mb + tap / 3][tap % 3] ^ (unsigned)(g << 5)) + abuf * A_BYTES) + t * A_TERM);
    };
    auto w_addr = [&](unsigned buf, int t, int g, int nb) -> const u32x4 * {
        return reinterpret_cast<const u32x4 *>(ldsb + B_BASE + buf * B_BYTES + lane * 16 + (((t * KG + g) * NBT + nb0 + nb) * 1024));
    };

    f32x16 acc[MB][NB];
    // C/D layout: lane&31 = pixel, register r = channel 32*nb + (r&3) + 8*(r>>2) + 4*(lane>>5): quad i = r>>2 is one b128 of the bias
    auto acc_init = [&]() {
#pragma unroll
        for (int mb = 0; mb < MB; mb++) {
            if constexpr (BINIT) {
                unsigned bo = BIAS_BASE + (unsigned)(nb0 * 32 + 4 * kk) * 4;
                asm volatile("" : "+v"(bo));   // one read per quad, landing in the accumulator registers (no shared copy + v_mov)
#pragma unroll
                for (int nb = 0; nb < NB; nb++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const f32x4 bq = *reinterpret_cast<const f32x4 *>(ldsb + bo + (nb * 32 + 8 * i) * 4);
#pragma unroll
                        for (int e = 0; e < 4; e++) acc[mb][nb][4 * i + e] = bq[e];
                    }
            } else {
#pragma unroll
                for (int nb = 0; nb < NB; nb++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[mb][nb][r] = 0.0f;
            }
        }
    };
    // bias + LeakyReLU of one accumulator register; b = its bias (used by the 4-wave shapes only)
    auto act = [&](float a, float b) -> float {
        // fp16 weights (and the bias an 8-wave accumulator started at) are pre-scaled by a power of two: undo it, exactly
        const float s = FMT ? a * d.acc_scale : a;
        if constexpr (BINIT) return leaky_acc(s);
        else { const float sb = s + b; return fmaxf(sb, 0.1f * sb); }
    };
    // the same for a register quad.  (Packed v_pk_mul_f32 for the two multiplies of an element pair measured 0.5 % slower on the one-term
    // shapes and equal on the two-term ones: packed fp32 VALU beside the partner wave's MFMAs is no gain.)
    auto act4 = [&](float a0, float a1, float a2, float a3, const f32x4 &bq, float (&v)[4]) {
        v[0] = act(a0, bq[0]); v[1] = act(a1, bq[1]); v[2] = act(a2, bq[2]); v[3] = act(a3, bq[3]);
    };
    // 4-wave shapes: the bias quads of this wave's planes live in registers for the whole kernel (up to 512 per wave there).  Reading
    // them from LDS in the epilogue put an exposed ds_read -> s_waitcnt round trip in front of every quad: 3-4k cycles per tile,
    // 42 % of the one-term 32->64 tile (s_memtime, round 2).
    f32x4 bqr[BINIT ? 1 : NB][4];
    if constexpr (!BINIT) {
#pragma unroll
        for (int nb = 0; nb < NB; nb++)
#pragma unroll
            for (int i = 0; i < 4; i++) bqr[nb][i] = *reinterpret_cast<const f32x4 *>(d.bias + (nb0 + nb) * 32 + 8 * i + 4 * kk);
    }

    // ---- prologue: A(slice 0) and B stages 0..LEAD-1 of the first tile ----
    tile_offsets(tile);
#pragma unroll
    for (int jj = 0; jj < APW; jj++) dma_a(0, 0, jj);
#pragma unroll
    for (int t = 0; t < LEAD; t++)
#pragma unroll
        for (int jb = 0; jb < BPW; jb++) dma_b(0, t, t, jb);
    W2XC_WAIT_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    acc_init();

    unsigned gs = 0;      // ring slot of the current stage
    unsigned abuf = 0;    // A buffer of the current slice
    int sl = 0;
    bool epi_stores = false;   // an interior-tile epilogue (NST stores) directly precedes the current slice
    u32x4 x_cur[T][MB], w_cur[T][NB];
#pragma unroll
    for (int t = 0; t < T; t++) {   // (the two-term schedule reads x0 / w1 inside the step)
#pragma unroll
        for (int mb = 0; mb < MB; mb++) x_cur[t][mb] = *x_addr(0, (T == 2) ? 1 : t, mb, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; nb++) w_cur[t][nb] = *w_addr(0, (T == 2) ? 0 : t, 0, nb);
    }

    for (;;) {
        // what the A pieces issued during this slice fetch: the next slice, or the next tile's first
        const bool last_slice = (sl == NSL - 1);
        unsigned a_add = (unsigned)(sl + 1) * KG * gs16;
        if (last_slice) {
            tile_offsets(tile + per < chunk_end ? tile + per : tile);
            a_add = 0;
        }
        const int sl_next = last_slice ? 0 : sl + 1;

        static_for<0, 9>([&](auto TAP) {
            constexpr int tap = decltype(TAP)::value;
            constexpr int tapL = (tap + LEAD) % 9;
            const int slL = (tap + LEAD < 9) ? sl : sl_next;
            const unsigned buf = gs, bufL = gs + LEAD >= RING ? gs + LEAD - RING : gs + LEAD, buf1 = gs + 1 >= RING ? 0 : gs + 1;
            // A pieces of the next slice: ka(t) = ceil-spread of APW over taps 0..LASTA, at most 2 per tap
            constexpr int ja0 = ka_before<APW, LASTA>(tap);
            if constexpr (T == 2) {
                // Two terms, one k-group per stage: "rolling" fragment registers.  Products run (x1,w0), (x0,w0), (x0,w1);
                // each fragment group is reloaded right after its last use and first needed >= MB*NB MFMAs later:
                //   product 0 window: x0, w1 of THIS step      (free since the previous step's last product)
                //   product 1 window: the step's DMAs, then x1 of the NEXT step
                //   product 2 window: w0 of the NEXT step
                // so one set of fragment registers (12 for the 4x2 blocking) serves instead of two.
                static_assert(KG == 1, "rolling schedule: one k-group per stage");
                constexpr int Q = MB * NB, M = 3 * Q;
                constexpr int n_a = ka<APW, LASTA>(tap), n_b = BPW;
                constexpr int tap_n = (tap + 1) % 9;
                const unsigned abuf_n = (tap == 8) ? (abuf ^ 1u) : abuf;
                static_for<0, M>([&](auto MI) {
                    constexpr int m = decltype(MI)::value;
                    constexpr int pi = m / Q, j = m % Q, mb = j / NB, nb = j % NB;
                    if constexpr (FMT == 1)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            __builtin_bit_cast(h16x8, w_cur[Prod<2>::b(pi)][nb]),
                            __builtin_bit_cast(h16x8, x_cur[Prod<2>::a(pi)][mb]), acc[mb][nb], 0, 0, 0);
                    else
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, w_cur[Prod<2>::b(pi)][nb]),
                            __builtin_bit_cast(bf16x8, x_cur[Prod<2>::a(pi)][mb]), acc[mb][nb], 0, 0, 0);
                    // fillers of this window that sit behind MFMA j: windows 0 and 2 front-loaded, window 1 spread
                    constexpr int nwin = pi == 0 ? MB + NB : pi == 1 ? n_a + n_b + MB : NB;
                    static_for<0, nwin>([&](auto FI) {
                        constexpr int f = decltype(FI)::value;
                        constexpr int slot = pi == 1 ? (f * Q) / nwin : (f < Q ? f : Q - 1);
                        if constexpr (slot == j) {
                            __builtin_amdgcn_sched_barrier(0);
                            if constexpr (pi == 0) {
                                if constexpr (f < MB) x_cur[0][f] = *x_addr(abuf, 0, f, tap, 0);
                                else w_cur[1][f - MB] = *w_addr(buf, 1, 0, f - MB);
                            } else if constexpr (pi == 1) {
                                if constexpr (f < n_a) dma_a(a_add, abuf ^ 1u, ja0 + f);
                                else if constexpr (f < n_a + n_b) dma_b(slL, tapL, bufL, f - n_a);
                                else x_cur[1][f - n_a - n_b] = *x_addr(abuf_n, 1, f - n_a - n_b, tap_n, 0);
                            } else {
                                w_cur[0][f] = *w_addr(buf1, 0, 0, f);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    });
                });
                __builtin_amdgcn_sched_barrier(0);
            } else
            static_for<0, KG>([&](auto G) {
                constexpr int g = decltype(G)::value;
                // One step = M MFMAs on k-group g; the other instructions are pinned into MFMA shadows:
                // the DMAs of this step first, then the fragment reads of the NEXT step (the last step of a
                // stage reads the next stage's first fragments, complete since the previous barrier).
                // Filler f sits behind MFMA (f * D) / F, D = M - LATE: the last LATE MFMAs cover the LDS latency.
                constexpr int M = NP * MB * NB, R = T * (MB + NB);
                constexpr int n_a = (g == 0) ? ka<APW, LASTA>(tap) : 0;
                constexpr int n_b = g < BPW ? (BPW - g + KG - 1) / KG : 0;
                constexpr int F = n_a + n_b + R;
                constexpr int LATE = W2XC_SPLIT_LATE < M / 2 ? W2XC_SPLIT_LATE : M / 2;
                constexpr int D = M - LATE;
                constexpr bool wrap = (g == KG - 1);
                constexpr int tap_n = wrap ? (tap + 1) % 9 : tap, g_n = wrap ? 0 : g + 1;
                const unsigned abuf_n = (wrap && tap == 8) ? (abuf ^ 1u) : abuf;
                const unsigned bbuf_n = wrap ? buf1 : buf;
                u32x4 x_nxt[T][MB], w_nxt[T][NB];
                static_for<0, M>([&](auto MI) {
                    constexpr int m = decltype(MI)::value;            // MFMA index in the step
                    constexpr int pi = m / (MB * NB), mb = (m / NB) % MB, nb = m % NB;
                    if constexpr (FMT == 1)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            __builtin_bit_cast(h16x8, w_cur[Prod<T>::b(pi)][nb]),
                            __builtin_bit_cast(h16x8, x_cur[Prod<T>::a(pi)][mb]), acc[mb][nb], 0, 0, 0);
                    else
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, w_cur[Prod<T>::b(pi)][nb]),
                            __builtin_bit_cast(bf16x8, x_cur[Prod<T>::a(pi)][mb]), acc[mb][nb], 0, 0, 0);
                    constexpr int f0 = m < D ? (m * F + D - 1) / D : F;
                    constexpr int f1 = m < D ? (((m + 1) * F + D - 1) / D < F ? ((m + 1) * F + D - 1) / D : F) : F;
                    static_for<f0, f1>([&](auto FI) {
                        constexpr int f = decltype(FI)::value;
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (f < n_a) {
                            dma_a(a_add, abuf ^ 1u, ja0 + f);
                        } else if constexpr (f < n_a + n_b) {
                            dma_b(slL, tapL, bufL, g + (f - n_a) * KG);
                        } else {
                            constexpr int code = read_decode<T, MB, NB>(f - n_a - n_b);
                            constexpr int t = (code / 10) % 10, u = code % 10;
                            if constexpr (code < 100) x_nxt[t][u] = *x_addr(abuf_n, t, u, tap_n, g_n);
                            else w_nxt[t][u] = *w_addr(bbuf_n, t, g_n, u);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < T; t++) {
#pragma unroll
                    for (int mb = 0; mb < MB; mb++) x_cur[t][mb] = x_nxt[t][mb];
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) w_cur[t][nb] = w_nxt[t][nb];
                }
            });
            // stage boundary: B(t+2) -- issued LOOK stages ago -- must have landed; the only younger transfers
            // are the ones issued since (vmcnt retires in order).  For the first LOOK stages after an interior
            // epilogue the NST stores are younger than B(t+2) too and are counted in rather than drained.
            if constexpr ((tap + 1) % E == 0) {
                constexpr int n_dma = ka_window<APW, LASTA>(tap, LOOK) + LOOK * BPW;
                if (tap < LOOK && sl == 0 && epi_stores) wait_vmcnt_n(n_dma + NST);
                else wait_vmcnt_n(n_dma);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            gs = buf1;
        });
        abuf ^= 1u;
        epi_stores = false;

        if (last_slice) {
            // ---- epilogue: bias + LeakyReLU in fp32, split into OT bf16 terms (or fp32), NHWC stores.
            //      C/D: lane&31 = pixel, channel = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
            const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
            const int oy0 = tile_y * ROWS, ox0 = tile_x * 32;
            const bool interior = (oy0 + ROWS <= d.out_h) && (ox0 + 32 <= d.out_w);
            // fp32 out: NHWC.  Term planes: channel-group blocked, element (t, c, y, x) at
            //   t*ts + (c / 16)*gs + y*rs + x*16 + c % 16   (the layout the next layer's A tiles stream from)
            const long long obase = OT == 0 ? (long long)(oy0 + wm * MB) * d.out_rs + (long long)(ox0 + li) * COUT + nb0 * 32 + 4 * kk
                                            : (long long)(oy0 + wm * MB) * d.out_rs + (long long)(ox0 + li) * GRP;
            constexpr int OES = OT == 0 ? 4 : 2;                                              // bytes per stored element
            const unsigned lane_ob = (unsigned)(OT == 0 ? li * COUT + 4 * kk : li * GRP + 4 * kk) * OES;   // this lane's byte offset inside the tile
            auto oofs = [&](int mb, int nb, int i) -> long long {   // element offset of channels (nb0+nb)*32 + 8i + 4kk .. +3
                if (OT == 0) return obase + (long long)mb * d.out_rs + nb * 32 + 8 * i;
                // (GRP = 16: the channel group (nb0 + nb) * 2 + i / 2 is wave-uniform -- written as c / GRP with the lane's kk inside c, its product with out_gs
                //  became a 64-bit per-lane value per (nb, i), hoisted out of the tile loop and spilled: 34 registers in the 128 -> 128 bf16 kernel, round 6)
                static_assert(GRP == 16, "channel groups of 16");
                const int cg = (nb0 + nb) * 2 + (i >> 1), cl = 8 * (i & 1) + 4 * kk;
                return obase + (long long)mb * d.out_rs + (long long)cg * d.out_gs + cl;
            };
            if constexpr (OT == 9) {
                // ---- the last layer (cin = COUT -> 1 plane, 3x3) inside this epilogue, "taps as rows":
                //   G[tap][pixel] = sum over this wave's channels of W7[tap][c] * act[c][pixel]  on the same 16-bit MFMA.
                // The accumulator registers already ARE a B operand: lane (pixel, kk), registers 8h .. 8h+7 of a plane
                // block hold 8 of the 16 channels of k-group h (16h + 4kk + {0..3, 8..11}); the weight fragments were
                // packed with the same channel order.  Activations are split into two terms exactly like stored
                // ones (one term in the one-term mode: W2XC_PRECISION_BF16 means bf16 operands everywhere); same product lists as the
                // main loop.  Each wave column (wn) writes its partial G as 9 tap planes
                // [half][tap][y][x] (128-byte runs per store) and conv3x3_last_gather adds the halves and the taps.
                u32x4 w7[LT][NB][2];
#pragma unroll
                for (int t = 0; t < LT; t++)
#pragma unroll
                    for (int nb = 0; nb < NB; nb++)
#pragma unroll
                        for (int h = 0; h < 2; h++)
                            w7[t][nb][h] = W7_IN_LDS ? *reinterpret_cast<const u32x4 *>(ldsb + W7_BASE + ((((t * NBT + nb0 + nb) * 2 + h) * 64 + lane) * 16))
                                                     : reinterpret_cast<const u32x4 *>(d.w7pk)[((t * NBT + nb0 + nb) * 2 + h) * 64 + lane];
                const bool xin = ox0 + li < d.out_w;
                float *gbase = d.out + (long long)wn * d.out_ts + (long long)(4 * kk) * d.out_gs + (long long)(oy0 + wm * MB) * d.out_rs + (ox0 + li);
#pragma unroll
                for (int mb = 0; mb < MB; mb++) {
                    f32x16 g;
#pragma unroll
                    for (int r = 0; r < 16; r++) g[r] = 0.0f;
#pragma unroll
                    for (int nb = 0; nb < NB; nb++)
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            float a[8];
#pragma unroll
                            for (int q = 0; q < 2; q++) {
                                const f32x4 bq = bqr[BINIT ? 0 : nb][2 * h + q];   // (unused by the 8-wave shapes)
                                float v4[4];
                                act4(acc[mb][nb][8 * h + 4 * q], acc[mb][nb][8 * h + 4 * q + 1], acc[mb][nb][8 * h + 4 * q + 2], acc[mb][nb][8 * h + 4 * q + 3], bq, v4);
#pragma unroll
                                for (int e = 0; e < 4; e++) a[4 * q + e] = FMT ? __builtin_amdgcn_fmed3f(v4[e], -65504.0f, 65504.0f) : v4[e];
                            }
                            u32x4 xt[LT];   // activation terms of this k-group as B operands
#pragma unroll
                            for (int t = 0; t < LT; t++)
#pragma unroll
                                for (int u = 0; u < 4; u++) {
                                    const unsigned ph = FMT ? pk_f16(a[2 * u], a[2 * u + 1]) : pk_bf16(a[2 * u], a[2 * u + 1]);
                                    xt[t][u] = ph;
                                    if (t + 1 < LT) {
                                        if (FMT) {
                                            a[2 * u] = sub_f16<0>(a[2 * u], ph);
                                            a[2 * u + 1] = sub_f16<1>(a[2 * u + 1], ph);
                                        } else {
                                            a[2 * u] -= __uint_as_float(ph << 16);
                                            a[2 * u + 1] -= __uint_as_float(ph & 0xFFFF0000u);
                                        }
                                    }
                                }
                            static_for<0, Prod<LT>::N>([&](auto PI) {
                                constexpr int pi = decltype(PI)::value;
                                if constexpr (FMT == 1)
                                    g = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, w7[Prod<LT>::b(pi)][nb][h]),
                                                                               __builtin_bit_cast(h16x8, xt[Prod<LT>::a(pi)]), g, 0, 0, 0);
                                else
                                    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w7[Prod<LT>::b(pi)][nb][h]),
                                                                                __builtin_bit_cast(bf16x8, xt[Prod<LT>::a(pi)]), g, 0, 0, 0);
                            });
                        }
                    // rows of g held by lane (pixel, kk): registers 0..3 = taps 4kk .. 4kk+3, register 4 = tap 8 (kk = 0)
                    const float gs = FMT ? d.g_scale : 1.0f;
                    float *gp = gbase + (long long)mb * d.out_rs;
                    if (interior || (xin && (oy0 + wm * MB + mb < d.out_h))) {
#pragma unroll
                        for (int e = 0; e < 4; e++) gp[(long long)e * d.out_gs] = g[e] * gs;
                        if (kk == 0) gp[8 * d.out_gs] = g[4] * gs;
                    }
                }
                epi_stores = interior;
            } else if (interior) {
                gbyte_t *lp = nullptr;
                static_for<0, MB * NB * 4>([&](auto IDX) {
                            constexpr int mb = decltype(IDX)::value / (NB * 4), nb = (decltype(IDX)::value / 4) % NB, i = decltype(IDX)::value % 4;
                            float v[4];
                            const f32x4 bq = bqr[BINIT ? 0 : nb][i];   // (unused by the 8-wave shapes)
                            act4(acc[mb][nb][4 * i], acc[mb][nb][4 * i + 1], acc[mb][nb][4 * i + 2], acc[mb][nb][4 * i + 3], bq, v);
                            if constexpr (NW == 8) {
                                // 256 registers per wave: a wave-uniform 64-bit base from the scalar ALU per (row, 16-plane group) + one lane
                                // offset, instead of 64-bit per-lane bases that spill (4-wave shapes: per-lane bases + immediates, as before)
                                constexpr bool newbase = OT == 0 ? (i == 0) : ((i & 1) == 0);
                                if constexpr (newbase) {
                                    const long long uo = OT == 0 ? (long long)(oy0 + wm * MB + mb) * d.out_rs + (long long)ox0 * COUT + (nb0 + nb) * 32
                                                                 : (long long)(oy0 + wm * MB + mb) * d.out_rs + (long long)ox0 * GRP +
                                                                       (long long)((nb0 + nb) * 2 + (i >> 1)) * d.out_gs;
                                    gbyte_t *ub = (gbyte_t *)(reinterpret_cast<char *>(d.out)) + uo * OES;
                                    asm volatile("" : "+s"(ub));   // (keeps the scalar base apart from the lane offset through instruction selection)
                                    lp = ub + lane_ob;
                                }
                                store_terms_at<OT, FMT>(lp + (OT == 0 ? 32 * i : 16 * (i & 1)), d.out_ts * OES, v[0], v[1], v[2], v[3]);
                            } else {
                                store_terms<OT, FMT>(d.out, oofs(mb, nb, i), d.out_ts, v[0], v[1], v[2], v[3]);
                            }
                });
                epi_stores = true;
            } else {
                const bool xin = ox0 + li < d.out_w;
#pragma unroll
                for (int mb = 0; mb < MB; mb++) {
                    const bool in = xin && (oy0 + wm * MB + mb < d.out_h);
#pragma unroll
                    for (int nb = 0; nb < NB; nb++)
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            float v[4];
                            const f32x4 bq = bqr[BINIT ? 0 : nb][i];   // (unused by the 8-wave shapes)
                            act4(acc[mb][nb][4 * i], acc[mb][nb][4 * i + 1], acc[mb][nb][4 * i + 2], acc[mb][nb][4 * i + 3], bq, v);
                            if (in) store_terms<OT, FMT>(d.out, oofs(mb, nb, i), d.out_ts, v[0], v[1], v[2], v[3]);
                        }
                }
            }
            tile += per;
            if (tile >= chunk_end) break;
            acc_init();
            sl = 0;
        } else {
            sl++;
        }
    }
    W2XC_WAIT_VMCNT(0);   // drain the speculative DMAs before the LDS is released
}

// ------------------------------------------------------------------------------------------------
// conv3x3_first_split: conv3x3_first (K = 9*CIN on v_mfma_f32_32x32x2_f32, fp32 planar input with the
// clamp-to-edge pad and the optional nearest-neighbour 2x folded into the LDS fill) with the operands
// swapped, so that a lane holds 4 consecutive channels of one pixel and stores OT term planes directly.
// ------------------------------------------------------------------------------------------------
template <int CIN, int NBT, int OT, int FMT>
__global__ void __launch_bounds__(256) conv3x3_first_split(W2xcConvDesc d, int tiles_x, int ntiles)
{
    constexpr int ROWS = 8, MB = 2, HW = 34, HH = ROWS + 2;
    constexpr int OSLC = 16;                          // channel-group size of the consumer (16 * its KG)
    constexpr int K = 9 * CIN, S = (K + 1) / 2;
    constexpr int COUT = 32 * NBT;
    __shared__ float lds[CIN * HH * HW];

    const int tile = xcd_remap(blockIdx.x, ntiles);
    const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
    const int oy0 = tile_y * ROWS, ox0 = tile_x * 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    for (int idx = threadIdx.x; idx < CIN * HH * HW; idx += 256) {
        const int c = idx / (HH * HW), p = idx - c * (HH * HW);
        const int py = p / HW, px = p - py * HW;
        const int gy = clampi(oy0 + py + d.off_y, 0, d.in_h - 1) >> d.in_shift;
        const int gx = clampi(ox0 + px + d.off_x, 0, d.in_w - 1) >> d.in_shift;
        lds[idx] = d.in[(long long)c * d.in_cs + (long long)gy * d.in_rs + (long long)gx * d.in_ps];
    }
    __syncthreads();

    const int kk = lane >> 5, i = lane & 31;
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

    const int x = ox0 + i;
#pragma unroll 1
    for (int nb = 0; nb < NBT; nb++) {
        float b[S];
#pragma unroll
        for (int s = 0; s < S; s++) b[s] = d.wpk[(nb * S + s) * 64 + lane];
        f32x16 acc[MB];
#pragma unroll
        for (int mb = 0; mb < MB; mb++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mb][r] = 0.0f;
#pragma unroll
        for (int s = 0; s < S; s++)
#pragma unroll
            for (int mb = 0; mb < MB; mb++)
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s], a[mb][s], acc[mb], 0, 0, 0);   // rows = channels, columns = pixels
#pragma unroll
        for (int mb = 0; mb < MB; mb++) {
            const int y = oy0 + wave * MB + mb;
            if (y < d.out_h && x < d.out_w) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = leaky(acc[mb][4 * q + e] + d.bias[nb * 32 + 8 * q + 4 * kk + e]);
                    const int c = nb * 32 + 8 * q + 4 * kk;      // blocked term planes (see conv3x3_split)
                    store_terms<OT, FMT>(d.out, (long long)(c / OSLC) * d.out_gs + (long long)y * d.out_rs + (long long)x * OSLC + c % OSLC, d.out_ts,
                                    v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// conv3x3_first2_split: layers 1 AND 2 in one kernel (layer 1 = 1 -> 32 planes, layer 2 = 32 -> COUT), the layer-1
// activations never leave the CU.  Tile = 8 rows x 32 px of layer 2's output, 4 waves:
//   1. the 12 x 36 input patch (clamp-to-edge pad and the optional nearest-2x folded in, like conv3x3_first) -> LDS;
//   2. layer 1 on the 10 x 34 halo pixels, 32 at a time, on v_mfma_f32_32x32x2_f32 (operands swapped: lane = pixel),
//      bias + LeakyReLU, split into T 16-bit terms, written to LDS as [term][pixel][32 channels] (64 B per pixel,
//      16-byte chunks XOR-swizzled with bits 2..3 of the pixel index: conflict-free ds_read_b128);
//   3. layer 2 from those terms exactly like conv3x3_split (same product order, same stage order: bit-identical
//      results), weight fragments straight from L2 in w2xc_split_pack order, several workgroups per CU hide latency.
// Saves layer 1's store and layer 2's load (128 / 192 B per pixel at two / three terms) and one launch.
// ------------------------------------------------------------------------------------------------
template <int COUT, int T, int OT, int FMT>
__global__ void __launch_bounds__(256) conv3x3_first2_split(W2xcConvDesc d, int tiles_x, int ntiles)
{
    constexpr int ROWS = 8, MB = 2, HW = 34, HH = ROWS + 2, NPIX = HH * HW, NBLK = (NPIX + 31) / 32;
    constexpr int PW = 36, PH = HH + 2;
    constexpr int NBT = COUT / 32, S = 5;                       // layer 1: K = 9 taps padded to 10 = 5 MFMA k-steps
    constexpr int KGX = (T == 1) ? 2 : 1, NSLX = 2 / KGX;      // structure of layer 2's weight image (w2xc_split_pack, cin = 32)
    constexpr int NP = Prod<T>::N;
    constexpr unsigned ACT_BASE = 2048, ACT_TERM = NBLK * 32 * 64;
    static_assert(PH * PW * 4 <= ACT_BASE && (OT == T || OT == 0), "shape");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char *ldsb = reinterpret_cast<char *>(lds);

    const int tile = xcd_remap(blockIdx.x, ntiles);
    const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
    const int oy0 = tile_y * ROWS, ox0 = tile_x * 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kk = lane >> 5, li = lane & 31;

    // ---- 1. input patch (every pass's load in flight at once: as a loop it was load / s_waitcnt vmcnt(0) / write per pass) ----
    {
        constexpr int PP = (PH * PW + 255) / 256;
        float pv[PP];
#pragma unroll
        for (int t = 0; t < PP; t++) {
            const int idx = min((int)threadIdx.x + 256 * t, PH * PW - 1);
            const int py = idx / PW, px = idx - py * PW;
            const int gy = clampi(oy0 + py + d.off_y, 0, d.in_h - 1) >> d.in_shift;
            const int gx = clampi(ox0 + px + d.off_x, 0, d.in_w - 1) >> d.in_shift;
            pv[t] = d.in[(long long)gy * d.in_rs + (long long)gx * d.in_ps];
        }
#pragma unroll
        for (int t = 0; t < PP; t++)
            if ((int)threadIdx.x + 256 * t < PH * PW) lds[threadIdx.x + 256 * t] = pv[t];
    }
    float w1[S], b1[16];
#pragma unroll
    for (int s = 0; s < S; s++) w1[s] = d.w1pk[s * 64 + lane];
#pragma unroll
    for (int r = 0; r < 16; r++) b1[r] = d.bias1[(r & 3) + 8 * (r >> 2) + 4 * kk];
    __syncthreads();

    // ---- 2. layer 1 on the halo tile -> term planes in LDS ----
    for (int blk = wave; blk < NBLK; blk += 4) {
        const int p = blk * 32 + li;
        const int pc = p < NPIX ? p : NPIX - 1;
        const int py = pc / HW, px = pc - py * HW;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < S; s++) {
            const int k0 = 2 * s, k1 = 2 * s + 1;
            const int off0 = (k0 / 3) * PW + k0 % 3, off1 = k1 < 9 ? (k1 / 3) * PW + k1 % 3 : 0;
            const float a = lds[py * PW + px + (kk ? off1 : off0)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[s], a, acc, 0, 0, 0);   // rows = channels, columns = pixels
        }
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            v[r] = leaky(acc[r] + b1[r]);
            if (FMT) v[r] = __builtin_amdgcn_fmed3f(v[r], -65504.0f, 65504.0f);
        }
        // registers 8g + 4h + (0..3) = channels 16g + 8h + 4kk + (0..3): the kk-th 8-byte half of chunk (g, h)
        const unsigned pbase = ACT_BASE + (unsigned)p * 64 + (unsigned)kk * 8;
        const unsigned sw = (unsigned)(p >> 2) & 3u;
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int c4 = 0; c4 < 4; c4++) {
                float *q = &v[4 * c4];
                const unsigned p01 = FMT ? pk_f16(q[0], q[1]) : pk_bf16(q[0], q[1]), p23 = FMT ? pk_f16(q[2], q[3]) : pk_bf16(q[2], q[3]);
                *reinterpret_cast<u32x2 *>(ldsb + pbase + t * ACT_TERM + (((unsigned)c4 ^ sw) << 4)) = (u32x2){p01, p23};
                if (t + 1 < T) {
                    if (FMT) {
                        q[0] = sub_f16<0>(q[0], p01); q[1] = sub_f16<1>(q[1], p01); q[2] = sub_f16<0>(q[2], p23); q[3] = sub_f16<1>(q[3], p23);
                    } else {
                        q[0] -= __uint_as_float(p01 << 16);
                        q[1] -= __uint_as_float(p01 & 0xFFFF0000u);
                        q[2] -= __uint_as_float(p23 << 16);
                        q[3] -= __uint_as_float(p23 & 0xFFFF0000u);
                    }
                }
            }
    }
    __syncthreads();

    // ---- 3. layer 2 ----
    // same arithmetic as the conv3x3_split shape that runs this layer unfused (bit-identical results): its 8-wave shapes
    // (two terms, 64 / 128 planes) start the accumulators at the pre-scaled bias, the 4-wave ones add the bias in the epilogue
    constexpr bool BINIT = (T == 2 && COUT >= 64);
    f32x16 acc2[MB][NBT];
    {
        const float bmul = FMT ? 1.0f / d.acc_scale : 1.0f;
#pragma unroll
        for (int nb = 0; nb < NBT; nb++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                f32x4 bq = {0.0f, 0.0f, 0.0f, 0.0f};
                if constexpr (BINIT) bq = *reinterpret_cast<const f32x4 *>(d.bias + nb * 32 + 8 * i + 4 * kk);
#pragma unroll
                for (int mb = 0; mb < MB; mb++)
#pragma unroll
                    for (int e = 0; e < 4; e++) acc2[mb][nb][4 * i + e] = BINIT ? bq[e] * bmul : 0.0f;
            }
    }
    // 18 (k-group, tap) iterations in conv3x3_split's stage order; the weight fragments come straight from L2, loaded
    // PF iterations ahead into a rotating register queue (the loops are fully unrolled: all indices are constants)
    const u32x4 *w2 = reinterpret_cast<const u32x4 *>(d.wpk) + lane;
    constexpr int NIT = 18, PF = 4;
    // fragment byte address of (row = mb + ty, tx) for k-group 0; k-group G is the same XOR (G << 5) (chunk 2G + kk = kk ^ 2G)
    unsigned xa[MB + 2][3];
#pragma unroll
    for (int row = 0; row < MB + 2; row++)
#pragma unroll
        for (int tx = 0; tx < 3; tx++) {
            const int p = (wave * MB + row) * HW + li + tx;
            xa[row][tx] = ACT_BASE + (unsigned)p * 64 + ((((unsigned)(p >> 2) & 3u) ^ (unsigned)kk) << 4);
        }
    u32x4 wq[PF][T][NBT];
    auto load_w = [&](int it) {
        const int sl = it / (9 * KGX), tap = (it / KGX) % 9, g = it % KGX;
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int nb = 0; nb < NBT; nb++) wq[it % PF][t][nb] = w2[(size_t)((((tap * NSLX + sl) * T + t) * KGX + g) * NBT + nb) * 64];
    };
    constexpr int PFX = 2;                                    // activation fragments are read PFX iterations ahead
    u32x4 xq[PFX + 1][T][MB];
    auto load_x = [&](int it) {
        const int sl = it / (9 * KGX), tap = (it / KGX) % 9, g = it % KGX;
        const int G = sl * KGX + g;                          // 16-channel k-group of layer 2's input
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int mb = 0; mb < MB; mb++)
                xq[it % (PFX + 1)][t][mb] = *reinterpret_cast<const u32x4 *>(ldsb + (xa[mb + tap / 3][tap % 3] ^ (unsigned)(G << 5)) + t * ACT_TERM);
    };
#pragma unroll
    for (int it = 0; it < PF - 1; it++) load_w(it);
#pragma unroll
    for (int it = 0; it < PFX; it++) load_x(it);
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        if (it + PF - 1 < NIT) load_w(it + PF - 1);
        if (it + PFX < NIT) load_x(it + PFX);
        __builtin_amdgcn_sched_barrier(0);                   // keep the prefetches above this iteration's MFMAs
        auto &x = xq[it % (PFX + 1)];
#pragma unroll
        for (int pi = 0; pi < NP; pi++)
#pragma unroll
            for (int mb = 0; mb < MB; mb++)
#pragma unroll
                for (int nb = 0; nb < NBT; nb++) {
                    if constexpr (FMT == 1)
                        acc2[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, wq[it % PF][Prod<T>::b(pi)][nb]),
                                                                              __builtin_bit_cast(h16x8, x[Prod<T>::a(pi)][mb]), acc2[mb][nb], 0, 0, 0);
                    else
                        acc2[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wq[it % PF][Prod<T>::b(pi)][nb]),
                                                                               __builtin_bit_cast(bf16x8, x[Prod<T>::a(pi)][mb]), acc2[mb][nb], 0, 0, 0);
                }
    }

    // ---- epilogue of layer 2: bias + LeakyReLU, term planes (blocked) or fp32 NHWC ----
    // (a plane block's four bias quads are fetched once, in front of its stores: fetched per store, every store stood behind s_waitcnt vmcnt(0) for its bias --
    //  and with it for the store before; the channel group c / 16 = 2 nb + i / 2 is wave-uniform and written so: round 6, found in the ISA)
    const int x = ox0 + li;
#pragma unroll
    for (int nb = 0; nb < NBT; nb++) {
        f32x4 bqs[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            bqs[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (!BINIT) bqs[i] = *reinterpret_cast<const f32x4 *>(d.bias + nb * 32 + 8 * i + 4 * kk);
        }
#pragma unroll
        for (int mb = 0; mb < MB; mb++) {
            const int y = oy0 + wave * MB + mb;
            const bool in = (y < d.out_h) && (x < d.out_w);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int c = nb * 32 + 8 * i + 4 * kk;
                float v[4];
                const f32x4 bq = bqs[i];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float sacc = FMT ? acc2[mb][nb][4 * i + e] * d.acc_scale : acc2[mb][nb][4 * i + e];
                    if constexpr (BINIT) v[e] = leaky_acc(sacc);
                    else { const float sb = sacc + bq[e]; v[e] = fmaxf(sb, 0.1f * sb); }
                }
                const long long o = OT == 0 ? (long long)y * d.out_rs + (long long)x * COUT + c
                                            : (long long)(nb * 2 + (i >> 1)) * d.out_gs + (long long)y * d.out_rs + (long long)x * 16 + (8 * (i & 1) + 4 * kk);
                if (in) store_terms<OT, FMT>(d.out, o, d.out_ts, v[0], v[1], v[2], v[3]);
            }
        }
    }
}

// ================================================================================================
// host side
// ================================================================================================
#if W2XC_SPLIT_T == 3   // shared host code lives in one object
// ---- last layer fused into the epilogue of the two-term kernels (out_terms = 9) ----
// conv3x3_last_gather: out(y,x) = leaky(bias + sum over taps (ty,tx) and wave-column halves of G[half][tap][y+ty][x+tx]),
// G = [halves][9][gh][gw] fp32 tap planes as written by conv3x3_split<.., OT = 9>; (gh, gw) = (out_h + 2, out_w + 2).
__global__ void __launch_bounds__(256) conv3x3_last_gather(const float *G, int halves, long long hs, long long ps, long long rs, long long xs,
                                                           const float *bias, float *out, long long out_rs, long long out_ps, int out_h, int out_w)
{
    const long long total = (long long)out_h * out_w;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int y = (int)(idx / out_w), x = (int)(idx - (long long)y * out_w);
        float v = 0.0f;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const float *g = G + tap * ps + (long long)(y + tap / 3) * rs + (long long)(x + tap % 3) * xs;
            for (int hf = 0; hf < halves; hf++) v += g[hf * hs];
        }
        out[(long long)y * out_rs + (long long)x * out_ps] = leaky(v + bias[0]);
    }
}

// The same sum for PLANAR partial planes (pixel stride 1, what conv3x3_split<.., OT = 9> writes), four consecutive pixels per thread: 9 * halves 16-byte loads
// (dword-aligned: the tap's column shift) and one 16-byte store instead of 4x as many dword accesses.  A row's last 1-3 pixels take the scalar route.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
__global__ void __launch_bounds__(256) conv3x3_last_gather_x4(const float *G, int halves, long long hs, long long ps, long long rs, const float *bias, float *out,
                                                              long long out_rs, int out_h, int out_w)
{
    const int gw = (out_w + 3) >> 2;
    const long long total = (long long)out_h * gw;
    const float b = bias[0];
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int y = (int)(idx / gw), x = (int)(idx - (long long)y * gw) * 4;
        if (x + 4 <= out_w) {
            f32x4u v = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                const float *g = G + tap * ps + (long long)(y + tap / 3) * rs + (x + tap % 3);
                for (int hf = 0; hf < halves; hf++) v += *reinterpret_cast<const f32x4u *>(g + hf * hs);
            }
            f32x4u o;
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = leaky(v[e] + b);
            *reinterpret_cast<f32x4u *>(out + (long long)y * out_rs + x) = o;
        } else {
            for (int xx = x; xx < out_w; xx++) {
                float v = 0.0f;
#pragma unroll
                for (int tap = 0; tap < 9; tap++) {
                    const float *g = G + tap * ps + (long long)(y + tap / 3) * rs + (xx + tap % 3);
                    for (int hf = 0; hf < halves; hf++) v += g[hf * hs];
                }
                out[(long long)y * out_rs + xx] = leaky(v + b);
            }
        }
    }
}

hipError_t w2xc_launch_last_gather(const W2xcConvDesc &d, hipStream_t stream)
{
    if (d.out_w <= 0 || d.out_h <= 0) return hipSuccess;
    const long long total = (long long)d.out_h * d.out_w;
    int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    if (d.in_ps <= 1 && d.out_ps == 1) {   // planar partial planes, contiguous output pixels: four pixels per thread (0.19 -> 0.11 ms per 2160x3840 frame)
        const long long groups = (long long)d.out_h * ((d.out_w + 3) >> 2);
        grid = (int)((groups + 255) / 256 < 65536 ? (groups + 255) / 256 : 65536);
        hipLaunchKernelGGL(conv3x3_last_gather_x4, dim3(grid), dim3(256), 0, stream, d.in, d.halves, d.in_ts, d.in_gs, d.in_rs, d.bias, d.out, d.out_rs, d.out_h, d.out_w);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(conv3x3_last_gather, dim3(grid), dim3(256), 0, stream, d.in, d.halves, d.in_ts, d.in_gs, d.in_rs, d.in_ps > 0 ? d.in_ps : 1, d.bias, d.out,
                       d.out_rs, d.out_ps, d.out_h, d.out_w);
    return hipGetLastError();
}

// wave columns (WN) of the two-term tile shape for `cout` planes = partial-G planes the fused epilogue writes
int w2xc_split_halves(int terms, int cout) { return terms == 2 ? (cout >= 64 ? 2 : 1) : (cout >= 128 ? 2 : 1); }

size_t w2xc_split_pack_last_bytes(int cin, int terms) { return (size_t)terms * (cin / 32) * 2 * 64 * 8 * 2; }

// w7pk[term < terms][plane block][k-group h][lane][8] = term of S * W[0][c][tap = lane & 31] (0 for taps >= 9), with
// c = 32*block + 16*h + 4*(lane>>5) + (e < 4 ? e : 4 + e)   -- the channel order of the accumulator registers 8h .. 8h+7.
// Same scale rule as w2xc_split_pack.  w is [1][cin][3][3].
float w2xc_split_pack_last(int cin, int terms, int fmt, const float *w, void *dst)
{
    auto bf = [](float f) -> unsigned short {
        unsigned u;
        memcpy(&u, &f, 4);
        u += 0x7FFFu + ((u >> 16) & 1u);
        return (unsigned short)(u >> 16);
    };
    auto bf2f = [](unsigned short h) -> float {
        const unsigned u = (unsigned)h << 16;
        float f;
        memcpy(&f, &u, 4);
        return f;
    };
    float scale = 1.0f;
    if (fmt == 1) {
        float mx = 0.0f;
        for (int i = 0; i < 9 * cin; i++) mx = fabsf(w[i]) > mx ? fabsf(w[i]) : mx;
        if (mx > 0.0f && mx < INFINITY) {
            int e = 0;
            frexpf(mx, &e);
            scale = ldexpf(1.0f, 15 - e);
        }
    }
    const int nbt = cin / 32;
    unsigned short *d16 = static_cast<unsigned short *>(dst);
    for (int nb = 0; nb < nbt; nb++)
        for (int h = 0; h < 2; h++)
            for (int lane = 0; lane < 64; lane++)
                for (int e = 0; e < 8; e++) {
                    const int tap = lane & 31, kk = lane >> 5;
                    const int c = 32 * nb + 16 * h + 4 * kk + (e < 4 ? e : 4 + e);
                    float r = tap < 9 ? w[(size_t)c * 9 + tap] * scale : 0.0f;
                    for (int t = 0; t < terms; t++) {
                        unsigned short hv;
                        float back;
                        if (fmt == 1) {
                            const _Float16 hf = (_Float16)r;
                            memcpy(&hv, &hf, 2);
                            back = (float)hf;
                        } else {
                            hv = bf(r);
                            back = bf2f(hv);
                        }
                        d16[((((size_t)t * nbt + nb) * 2 + h) * 64 + lane) * 8 + e] = hv;
                        r -= back;
                    }
                }
    return scale;
}

// k-groups (16-channel layout groups) per stage: one for the two/three-term modes; the one-term mode has a third of
// the MFMAs per byte and takes 64-channel stages to amortise the stage barrier
int w2xc_split_kg(int terms, int cin) { return terms == 1 ? (cin >= 64 ? 4 : 2) : 1; }

size_t w2xc_split_packed_bytes(int cin, int cout, int terms) { return (size_t)9 * cin * cout * 2 * terms; }

// wpk[tap][slice][term][g][nb][lane][8] (16-bit) = term `term` of S * W[32*nb + (lane&31)][slice*16*KG + 16*g + 8*(lane>>5) + e][tap]
// fmt 0: bf16 terms, S = 1.  fmt 1: fp16 terms, S = the power of two that puts max|W| into [2^14, 2^15): the low
// term of every weight down to 2^-17 max|W| is then a NORMAL fp16 number (22 significant bits in two terms).
// Returns S; the consumer multiplies its accumulators by 1/S (exact).
float w2xc_split_pack(int cin, int cout, int terms, int fmt, const float *w, void *dst)
{
    auto bf = [](float f) -> unsigned short {
        unsigned u;
        memcpy(&u, &f, 4);
        u += 0x7FFFu + ((u >> 16) & 1u);
        return (unsigned short)(u >> 16);
    };
    auto bf2f = [](unsigned short h) -> float {
        const unsigned u = (unsigned)h << 16;
        float f;
        memcpy(&f, &u, 4);
        return f;
    };
    float scale = 1.0f;
    if (fmt == 1) {
        float mx = 0.0f;
        for (size_t i = 0; i < (size_t)9 * cin * cout; i++) mx = fabsf(w[i]) > mx ? fabsf(w[i]) : mx;
        if (mx > 0.0f && mx < INFINITY) {
            int e = 0;
            frexpf(mx, &e);                 // mx = f * 2^e, f in [0.5, 1)
            scale = ldexpf(1.0f, 15 - e);   // mx * scale in [2^14, 2^15)
        }
    }
    const int kg = w2xc_split_kg(terms, cin), nsl = cin / (16 * kg), nbt = cout / 32;
    unsigned short *d16 = static_cast<unsigned short *>(dst);
    for (int tap = 0; tap < 9; tap++)
        for (int sl = 0; sl < nsl; sl++)
            for (int nb = 0; nb < nbt; nb++)
                for (int g = 0; g < kg; g++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int e = 0; e < 8; e++) {
                            const int o = nb * 32 + (lane & 31), c = sl * 16 * kg + 16 * g + 8 * (lane >> 5) + e;
                            float r = w[((size_t)o * cin + c) * 9 + tap] * scale;   // exact (power of two)
                            for (int t = 0; t < terms; t++) {
                                unsigned short h;
                                float back;
                                if (fmt == 1) {
                                    const _Float16 hf = (_Float16)r;                 // round to nearest even
                                    memcpy(&h, &hf, 2);
                                    back = (float)hf;
                                } else {
                                    h = bf(r);
                                    back = bf2f(h);
                                }
                                d16[((((((size_t)tap * nsl + sl) * terms + t) * kg + g) * nbt + nb) * 64 + lane) * 8 + e] = h;
                                r -= back;
                            }
                        }
    return scale;
}
#endif

template <int CIN, int COUT, int MB, int NB, int WM, int WN, int T, int OT, int KG, int RING, int FMT, int E = 1>
static hipError_t launch_split(const W2xcConvDesc &d, hipStream_t stream)
{
    constexpr int ROWS = MB * WM, NPIXP = ((ROWS + 2) * 34 + 31) / 32 * 32;
    const int tiles_x = (d.out_w + 31) / 32, tiles_y = (d.out_h + ROWS - 1) / ROWS;
    const int ntiles = tiles_x * tiles_y;
    constexpr int NW = WM * WN;
    constexpr int A_PIECES = T * NPIXP * 2 * KG / 64, APW = (A_PIECES + NW - 1) / NW;
    constexpr size_t lds_bytes = 2 * (size_t)(NW * APW * 1024) + (size_t)RING * (T * KG * (COUT / 32) * 1024) + COUT * 4 +
                                 ((OT == 9 && T == 2 && E == 1) ? 4 * (COUT / 32) * 1024 : (OT == 9 && T == 1) ? 2 * (COUT / 32) * 1024 : 0);
    static_assert(lds_bytes <= 160 * 1024, "LDS budget");
    auto kern = conv3x3_split<CIN, COUT, MB, NB, WM, WN, T, OT, KG, RING, FMT, E>;
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
    if (grid > ((ntiles + 7) & ~7)) grid = (ntiles + 7) & ~7;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds_bytes, stream, d, tiles_x, ntiles);
    return hipGetLastError();
}

// Tile shapes per (cin, cout).  16 rows x 32 px wherever the LDS allows: every weight stage streamed from L2 serves
// twice the pixels and the halo overhead drops from 1.33 to 1.20 -- these kernels run against the power limit (shader
// clock 1.5-1.8 GHz), so bytes moved per MFMA are what is left to save.  Two terms: 8 waves (two per SIMD, rolling
// fragment registers) for the 64/128-plane outputs, 4 waves for 32.  Three terms: 4 waves; the 128-plane outputs stay
// on 8-row tiles (A[2] of a 16-row tile + the weight ring would need 192 KiB).
template <int T, int OT, int FMT>
static hipError_t launch_split_t(const W2xcConvDesc &d, hipStream_t stream)
{
    constexpr bool BIG = (T == 2);
    if constexpr (T == 1) {
        // One term (32-cycle MFMAs, one product per operand pair): the fragment reads of an 8-row tile with 2x2 blocks draw
        // the LDS's whole 128 bytes per clock, so the tilings below were picked by measurement (round 2, same box, same run):
        //   32->64    16 rows, 4 waves owning all 64 planes of 4 rows each (4x2 blocks), 32-channel stages, ring of 6:
        //             0.56 -> 0.39 ms per 2160x3840 layer (a fragment read feeds 1.5x the MFMAs, weights stream once per 16 rows)
        //   128->128  16 rows, 8 waves (two per SIMD), 4x2 blocks, 32-channel stages: 2.40 -> 2.13 ms; and one workgroup barrier
        //             per 3 taps instead of per tap (E = 3, ring of 8): -> 2.02 ms.  The epoch schedule measured SLOWER on the
        //             other shapes (their 64-channel stages do not fit a ring of 8) and equal on the two-term kernels.
        //   64->64    8 rows, 4 waves, 64-channel stages, ring of 8 instead of 4: 0.74 -> 0.67 ms.  A stage of these kernels lasts
        //             0.5-1k cycles, so with a ring of 4 a weight stage is awaited ONE stage (< 0.5 us) after its issue -- less than
        //             an L2 round trip under load; 8 slots put 5 stages between issue and use.  (32->64 also runs a ring of 8.)
        //             Then 16 rows, 8 waves of 4 rows x 32 planes (4x1 blocks), 32-channel stages, ring of 8: 0.645 -> 0.60 ms (8 rows on
        //             8 waves of 2x1 blocks: the same).  One wave per SIMD hands the matrix pipe an MFMA every ~40 cycles, two every ~33
        //             (s_memtime); the same 8-wave shape on 32->64 measured equal to the 4-wave one (that layer is HBM-bound).
        //   64->128   8 rows, 8 waves (2x2 blocks), 32-channel stages, ring of 12 with one barrier per 3 taps (LOOK = 5 stages between
        //             a weight stage's issue and its use): 1.28 -> 1.16 ms.  (Its 64-channel stages are 16 KiB -- a ring of 4 is all
        //             that fits; the 16-row forms measured 4-15 % slower; 4x4 blocks for 128 planes spill ~290 registers.)
        //   the rest: 8 rows, 4 waves, 64-channel stages, ring of 4
        switch (d.cin * 1000 + d.cout) {
#ifndef W2XC_SPLIT_DEV
        case 32032:  return launch_split<32, 32, 2, 1, 4, 1, 1, OT, 2, 4, FMT>(d, stream);
        case 32064:  return launch_split<32, 64, 4, 2, 4, 1, 1, OT, 2, 8, FMT>(d, stream);
        case 32128:  return launch_split<32, 128, 4, 2, 2, 2, 1, OT, 2, 4, FMT>(d, stream);
        case 64032:  return launch_split<64, 32, 2, 1, 4, 1, 1, OT, 4, 4, FMT>(d, stream);
        case 64064:  return launch_split<64, 64, 4, 1, 4, 2, 1, OT, 2, 8, FMT>(d, stream);
        case 64128:  return launch_split<64, 128, 2, 2, 4, 2, 1, OT, 2, 12, FMT, 3>(d, stream);
        case 128032: return launch_split<128, 32, 2, 1, 4, 1, 1, OT, 4, 4, FMT>(d, stream);
        case 128064: return launch_split<128, 64, 2, 2, 4, 1, 1, OT, 4, 4, FMT>(d, stream);
#endif
        case 128128: return launch_split<128, 128, 4, 2, 4, 2, 1, OT, 2, 8, FMT, 3>(d, stream);
        default: return hipErrorInvalidValue;
        }
    }
    constexpr int KG = 1, RG = 6;
    switch (d.cin * 1000 + d.cout) {
#ifndef W2XC_SPLIT_DEV   // (development aid: -DW2XC_SPLIT_DEV instantiates 128->128 only)
    //                                     CIN  COUT  MB NB WM WN
    case 32032: { if constexpr (BIG) return launch_split<32, 32, 4, 1, 4, 1, T, OT, KG, RG, FMT>(d, stream); else return launch_split<32, 32, 4, 1, 4, 1, T, OT, KG, RG, FMT>(d, stream); }
    case 64032: { if constexpr (BIG) return launch_split<64, 32, 4, 1, 4, 1, T, OT, KG, RG, FMT>(d, stream); else return launch_split<64, 32, 4, 1, 4, 1, T, OT, KG, RG, FMT>(d, stream); }
    case 128032: { if constexpr (BIG) return launch_split<128, 32, 4, 1, 4, 1, T, OT, KG, RG, FMT>(d, stream); else return launch_split<128, 32, 4, 1, 4, 1, T, OT, KG, RG, FMT>(d, stream); }
    case 32064: { if constexpr (BIG) return launch_split<32, 64, 4, 1, 4, 2, T, OT, KG, RG, FMT>(d, stream); else return launch_split<32, 64, 4, 2, 4, 1, T, OT, KG, RG, FMT>(d, stream); }
    case 64064: { if constexpr (BIG) return launch_split<64, 64, 4, 1, 4, 2, T, OT, KG, RG, FMT>(d, stream); else return launch_split<64, 64, 4, 2, 4, 1, T, OT, KG, RG, FMT>(d, stream); }
    case 128064: { if constexpr (BIG) return launch_split<128, 64, 4, 1, 4, 2, T, OT, KG, RG, FMT>(d, stream); else return launch_split<128, 64, 4, 2, 4, 1, T, OT, KG, RG, FMT>(d, stream); }
    case 32128: { if constexpr (BIG) return launch_split<32, 128, 4, 2, 4, 2, T, OT, KG, RG, FMT>(d, stream); else return launch_split<32, 128, 4, 2, 2, 2, T, OT, KG, RG, FMT>(d, stream); }
    case 64128: { if constexpr (BIG) return launch_split<64, 128, 4, 2, 4, 2, T, OT, KG, RG, FMT>(d, stream); else return launch_split<64, 128, 4, 2, 2, 2, T, OT, KG, RG, FMT>(d, stream); }
#endif
    case 128128: { if constexpr (BIG) return launch_split<128, 128, 4, 2, 4, 2, T, OT, KG, RG, FMT>(d, stream); else return launch_split<128, 128, 4, 2, 2, 2, T, OT, KG, RG, FMT>(d, stream); }
    default: return hipErrorInvalidValue;
    }
}

template <int COUT, int T, int OT, int FMT>
static hipError_t launch_first2(const W2xcConvDesc &d, hipStream_t stream)
{
    const int tiles_x = (d.out_w + 31) / 32, tiles_y = (d.out_h + 7) / 8;
    const int ntiles = tiles_x * tiles_y;
    constexpr size_t lds_bytes = 2048 + (size_t)T * 11 * 32 * 64;
    auto kern = conv3x3_first2_split<COUT, T, OT, FMT>;
    if (lds_bytes > 64 * 1024) {   // function attributes are per device
        static std::atomic<unsigned long long> attr_done{0};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev >= 64 || !((attr_done.load() >> dev) & 1ull)) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return e;
            if (dev < 64) attr_done.fetch_or(1ull << dev);
        }
    }
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(256), lds_bytes, stream, d, tiles_x, ntiles);
    return hipGetLastError();
}

template <int T, int FMT>
static hipError_t launch_first2_t(const W2xcConvDesc &d, hipStream_t stream)
{
    if (d.cin != 32 || (d.out_terms != T && d.out_terms != 0)) return hipErrorInvalidValue;
    const bool f32out = d.out_terms == 0;
    switch (d.cout) {
    case 32:  return f32out ? launch_first2<32, T, 0, FMT>(d, stream) : launch_first2<32, T, T, FMT>(d, stream);
    case 64:  return f32out ? launch_first2<64, T, 0, FMT>(d, stream) : launch_first2<64, T, T, FMT>(d, stream);
    case 128: return f32out ? launch_first2<128, T, 0, FMT>(d, stream) : launch_first2<128, T, T, FMT>(d, stream);
    default: return hipErrorInvalidValue;
    }
}

template <int OT, int FMT>
static hipError_t launch_first_split_t(const W2xcConvDesc &d, hipStream_t stream)
{
    const int tiles_x = (d.out_w + 31) / 32, tiles_y = (d.out_h + 7) / 8;
    const int ntiles = tiles_x * tiles_y;
#define W2XC_FS(CIN, NBT) hipLaunchKernelGGL((conv3x3_first_split<CIN, NBT, OT, FMT>), dim3(ntiles), dim3(256), 0, stream, d, tiles_x, ntiles); break
    switch (d.cin * 1000 + d.cout) {
    case 1032:  W2XC_FS(1, 1);
    case 1064:  W2XC_FS(1, 2);
    case 1128:  W2XC_FS(1, 4);
    case 3032:  W2XC_FS(3, 1);
    case 3064:  W2XC_FS(3, 2);
    case 3128:  W2XC_FS(3, 4);
    default: return hipErrorInvalidValue;
    }
#undef W2XC_FS
    return hipGetLastError();
}

#if W2XC_SPLIT_T == 5   // three terms, fp32 / fused-last outputs (their own object: the six-product bodies compile slowly)
hipError_t w2xc_launch_split_mid_3x(const W2xcConvDesc &d, hipStream_t stream)
{
    return d.out_terms == 0 ? launch_split_t<3, 0, 0>(d, stream) : d.out_terms == 9 ? launch_split_t<3, 9, 0>(d, stream) : hipErrorInvalidValue;
}
#elif W2XC_SPLIT_T == 1
hipError_t w2xc_launch_split_mid_1(const W2xcConvDesc &d, hipStream_t stream)
{
    return d.out_terms == 1 ? launch_split_t<1, 1, 0>(d, stream) : d.out_terms == 0 ? launch_split_t<1, 0, 0>(d, stream)
         : d.out_terms == 9 ? launch_split_t<1, 9, 0>(d, stream) : hipErrorInvalidValue;
}
hipError_t w2xc_launch_split_first_1(const W2xcConvDesc &d, hipStream_t stream) { return launch_first_split_t<1, 0>(d, stream); }
hipError_t w2xc_launch_first2_1(const W2xcConvDesc &d, hipStream_t stream) { return launch_first2_t<1, 0>(d, stream); }
#elif W2XC_SPLIT_T == 2
hipError_t w2xc_launch_split_mid_2(const W2xcConvDesc &d, hipStream_t stream)
{
    return d.out_terms == 2 ? launch_split_t<2, 2, 0>(d, stream) : d.out_terms == 0 ? launch_split_t<2, 0, 0>(d, stream)
         : d.out_terms == 9 ? launch_split_t<2, 9, 0>(d, stream) : hipErrorInvalidValue;
}
hipError_t w2xc_launch_split_first_2(const W2xcConvDesc &d, hipStream_t stream) { return launch_first_split_t<2, 0>(d, stream); }
hipError_t w2xc_launch_first2_2(const W2xcConvDesc &d, hipStream_t stream) { return launch_first2_t<2, 0>(d, stream); }
#elif W2XC_SPLIT_T == 4   // fp16 x 2
hipError_t w2xc_launch_split_mid_h(const W2xcConvDesc &d, hipStream_t stream)
{
    return d.out_terms == 2 ? launch_split_t<2, 2, 1>(d, stream) : d.out_terms == 0 ? launch_split_t<2, 0, 1>(d, stream)
         : d.out_terms == 9 ? launch_split_t<2, 9, 1>(d, stream) : hipErrorInvalidValue;
}
hipError_t w2xc_launch_split_first_h(const W2xcConvDesc &d, hipStream_t stream) { return launch_first_split_t<2, 1>(d, stream); }
hipError_t w2xc_launch_first2_h(const W2xcConvDesc &d, hipStream_t stream) { return launch_first2_t<2, 1>(d, stream); }
#else
hipError_t w2xc_launch_split_mid_2(const W2xcConvDesc &d, hipStream_t stream);
hipError_t w2xc_launch_split_first_2(const W2xcConvDesc &d, hipStream_t stream);
hipError_t w2xc_launch_split_mid_h(const W2xcConvDesc &d, hipStream_t stream);
hipError_t w2xc_launch_split_first_h(const W2xcConvDesc &d, hipStream_t stream);
hipError_t w2xc_launch_split_mid_1(const W2xcConvDesc &d, hipStream_t stream);
hipError_t w2xc_launch_split_first_1(const W2xcConvDesc &d, hipStream_t stream);
hipError_t w2xc_launch_split_mid_3x(const W2xcConvDesc &d, hipStream_t stream);
hipError_t w2xc_launch_first2_1(const W2xcConvDesc &d, hipStream_t stream);
hipError_t w2xc_launch_first2_2(const W2xcConvDesc &d, hipStream_t stream);
hipError_t w2xc_launch_first2_h(const W2xcConvDesc &d, hipStream_t stream);

// layers 1 + 2 in one kernel (d.terms / d.fmt = format of the layer-1 activations, d.cin = 32)
hipError_t w2xc_launch_first2_split(const W2xcConvDesc &d, hipStream_t stream)
{
    if (d.out_w <= 0 || d.out_h <= 0) return hipSuccess;
    if (d.terms == 1 && d.fmt == 0) return w2xc_launch_first2_1(d, stream);
    if (d.terms == 2) return d.fmt == 1 ? w2xc_launch_first2_h(d, stream) : w2xc_launch_first2_2(d, stream);
    return (d.terms == 3 && d.fmt == 0) ? launch_first2_t<3, 0>(d, stream) : hipErrorInvalidValue;
}

hipError_t w2xc_launch_split_mid(const W2xcConvDesc &d, hipStream_t stream)
{
    if (d.out_w <= 0 || d.out_h <= 0) return hipSuccess;
    if (d.in_shift != 0 || (d.in_rs & 7) || (d.in_ts & 7) || (d.in_gs & 7)) return hipErrorInvalidValue;
    if (d.terms == 1 && d.fmt == 0) return w2xc_launch_split_mid_1(d, stream);
    if (d.terms == 2) return d.fmt == 1 ? w2xc_launch_split_mid_h(d, stream) : w2xc_launch_split_mid_2(d, stream);
    if (d.terms != 3 || d.fmt != 0) return hipErrorInvalidValue;
    return d.out_terms == 3 ? launch_split_t<3, 3, 0>(d, stream) : w2xc_launch_split_mid_3x(d, stream);
}

hipError_t w2xc_launch_split_first(const W2xcConvDesc &d, hipStream_t stream)
{
    if (d.out_w <= 0 || d.out_h <= 0) return hipSuccess;
    if (d.out_terms == 1 && d.fmt == 0) return w2xc_launch_split_first_1(d, stream);
    if (d.out_terms == 2) return d.fmt == 1 ? w2xc_launch_split_first_h(d, stream) : w2xc_launch_split_first_2(d, stream);
    return (d.out_terms == 3 && d.fmt == 0) ? launch_first_split_t<3, 0>(d, stream) : hipErrorInvalidValue;
}
#endif
