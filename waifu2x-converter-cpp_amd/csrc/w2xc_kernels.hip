// w2xc_kernels.hip -- gfx950 (MI355X, CDNA4) kernels for the waifu2x 3x3-conv + bias + LeakyReLU
// layer, i.e. the body of Model::filterWorker (/root/reference/src/modelHandler.cpp:117-159).
//
// Four kernels, one launch per layer per band (see w2xc_kernels.h for the addressing contract):
//
//   conv3x3_direct      any (cin, cout), any strides.  VALU, one thread per output pixel and 8
//                       output planes; sums in the REFERENCE'S order with unfused mul/add, so it
//                       is bit-exact against the CPU oracle.  Fallback + on-GPU cross-check.
//   conv3x3_mfma        cin, cout in {32,64,128}.  NHWC fp32 activations; the (rows+2) x 34 pixel
//                       halo tile of a 32-channel slice is staged in LDS (pixel stride 36 floats:
//                       conflict-free ds_read_b128), the 3x3xCinxCout contraction is an implicit
//                       GEMM on v_mfma_f32_32x32x2_f32 (M = 32 pixels of one row, N = 32 output
//                       planes, K = (tap, cin)), weights stream from L2 in fragment order
//                       (one coalesced dwordx4 per lane per 4 MFMAs), bias + LeakyReLU fused
//                       into the epilogue, NHWC stores in 128-byte runs.
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

#include <string.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static __device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// cv::max / cv::min / cv::scaleAdd(neg, 0.1, pos), modelHandler.cpp:148-152
static __device__ __forceinline__ float leaky(float v) { return v > 0.0f ? v : 0.1f * v; }

// XCD-aware block remap: the dispatcher places block b on XCD b % 8; give each XCD one contiguous
// range of tiles so neighbouring tiles (which share halo rows/columns) share an L2.  Bijective
// for any grid size (cdna_hip_programming.md T1).
static __device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

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
    for (int r = 0; r < 3; r++) roff[r] = (long long)clampi(y + r + d.off_y, 0, d.in_h - 1) * d.in_rs;
#pragma unroll
    for (int c = 0; c < 3; c++) coff[c] = (long long)clampi(x + c + d.off_x, 0, d.in_w - 1) * d.in_ps;
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
// conv3x3_mfma: fp32 MFMA implicit GEMM.
//   Workgroup = WM x WN waves; output tile = (MB*WM) rows x 32 pixels x COUT planes.
//   Wave (wm, wn) owns MB row-blocks (M = 32 pixels of one image row) x NB plane-blocks (N = 32).
//   K loop: cin slices of CC=32 staged in LDS; per slice 9 taps x 4 groups of 8 channels; per
//   group one ds_read_b128 per row-block (A) and one global dwordx4 per plane-block (B) feed
//   4 MFMA k-steps: lane (i = lane&31, kk = lane>>5) holds channels 4*kk..4*kk+3 of the group,
//   MFMA j contracts channels {j, 4+j}.
//   Packed weights: wpk[tap][cin/8][cout/32][lane][4], element j of lane (kk, n) =
//   W[o = 32*nb + n][i = 8*c8 + 4*kk + j][tap].
// ------------------------------------------------------------------------------------------------
template <int CIN, int COUT, int MB, int NB, int WM, int WN>
__global__ void __launch_bounds__(WM *WN * 64) conv3x3_mfma(W2xcConvDesc d, int tiles_x, int ntiles)
{
    constexpr int CC = 32;                 // channels per LDS slice
    constexpr int ROWS = MB * WM;
    constexpr int HW = 34, HH = ROWS + 2;  // halo tile
    constexpr int CS = CC + 4;             // LDS pixel stride (floats): 144 B, odd multiple of 16 B
    constexpr int NT = WM * WN * 64;
    constexpr int NBT = COUT / 32;
    static_assert(NB * WN == NBT, "plane blocks must tile COUT");
    static_assert(CIN % CC == 0, "cin must be a multiple of 32");
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tile = xcd_remap(blockIdx.x, ntiles);
    const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
    const int oy0 = tile_y * ROWS, ox0 = tile_x * 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; mb++)
#pragma unroll
        for (int nb = 0; nb < NB; nb++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mb][nb][r] = 0.0f;

    const float *a_lane = lds + ((wm * MB) * HW + (lane & 31)) * CS + (lane >> 5) * 4;
    const f32x4 *b_lane = reinterpret_cast<const f32x4 *>(d.wpk) + (wn * NB) * 64 + lane;

    // ---- halo-slice staging, split in two (issue early / write late): the global loads of slice
    //      c+1 are issued before the MFMA loop of slice c and land in registers while the matrix
    //      cores work; after the loop they are written to LDS between two barriers.  8 lanes move
    //      one pixel's 128 B.  Offsets are in float4 units (16 B) so 32 bits cover 64 GiB. ----
    constexpr int Q = CC / 4;
    constexpr int NFILL = HH * HW * Q;
    constexpr int PF = (NFILL + NT - 1) / NT;
    unsigned goff[PF];
    f32x4 pf[PF];
#pragma unroll
    for (int u = 0; u < PF; u++) {
        int idx = threadIdx.x + u * NT;
        idx = idx < NFILL ? idx : NFILL - 1;
        const int p = idx / Q, q = idx - p * Q;
        const int py = p / HW, px = p - py * HW;
        const int gy = clampi(oy0 + py + d.off_y, 0, d.in_h - 1);
        const int gx = clampi(ox0 + px + d.off_x, 0, d.in_w - 1);
        goff[u] = (unsigned)(((long long)gy * d.in_rs + (long long)gx * CIN) >> 2) + q;
    }
    const f32x4 *in4 = reinterpret_cast<const f32x4 *>(d.in);
#pragma unroll
    for (int u = 0; u < PF; u++) pf[u] = in4[goff[u]];

    for (int c0 = 0; c0 < CIN; c0 += CC) {
        if (c0) __syncthreads();   // every wave is done reading the previous slice
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int idx = threadIdx.x + u * NT;
            if (idx < NFILL) *reinterpret_cast<f32x4 *>(lds + (idx / Q) * CS + (idx % Q) * 4) = pf[u];
        }
        __syncthreads();
        if (c0 + CC < CIN) {
#pragma unroll
            for (int u = 0; u < PF; u++) pf[u] = in4[goff[u] + (c0 + CC) / 4];
        }

        const f32x4 *bp = b_lane + (long long)(c0 / 8) * NBT * 64;
        f32x4 a_cur[MB], b_cur[NB];
#pragma unroll
        for (int mb = 0; mb < MB; mb++) a_cur[mb] = *reinterpret_cast<const f32x4 *>(a_lane + (mb * HW) * CS);
#pragma unroll
        for (int nb = 0; nb < NB; nb++) b_cur[nb] = bp[nb * 64];
#pragma unroll
        for (int step = 0; step < 9 * (CC / 8); step++) {
            f32x4 a_nxt[MB], b_nxt[NB];
            if (step + 1 < 9 * (CC / 8)) {
                const int tap = (step + 1) / (CC / 8), c8 = (step + 1) % (CC / 8);
                const int ty = tap / 3, tx = tap % 3;
#pragma unroll
                for (int mb = 0; mb < MB; mb++)
                    a_nxt[mb] = *reinterpret_cast<const f32x4 *>(a_lane + ((mb + ty) * HW + tx) * CS + c8 * 8);
#pragma unroll
                for (int nb = 0; nb < NB; nb++)
                    b_nxt[nb] = bp[((long long)(tap * (CIN / 8) + c8) * NBT + nb) * 64];
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int mb = 0; mb < MB; mb++)
#pragma unroll
                    for (int nb = 0; nb < NB; nb++)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[mb][j], b_cur[nb][j],
                                                                          acc[mb][nb], 0, 0, 0);
            if (step + 1 < 9 * (CC / 8)) {
#pragma unroll
                for (int mb = 0; mb < MB; mb++) a_cur[mb] = a_nxt[mb];
#pragma unroll
                for (int nb = 0; nb < NB; nb++) b_cur[nb] = b_nxt[nb];
            }
        }
    }

    // ---- epilogue: bias + LeakyReLU, NHWC stores.  C/D map of 32x32 MFMA: column = lane&31
    //      (output plane), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel in the row-block). ----
#pragma unroll
    for (int mb = 0; mb < MB; mb++) {
        const int y = oy0 + wm * MB + mb;
        if (y >= d.out_h) continue;
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            const int n = (wn * NB + nb) * 32 + (lane & 31);
            const float bv = d.bias[n];
            float *orow = d.out + (long long)y * d.out_rs + n;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int x = ox0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (x < d.out_w) orow[(long long)x * COUT] = leaky(acc[mb][nb][r] + bv);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// conv3x3_first: cin <= 3, planar input, cout = 32*NBT planes, NHWC output.
//   K = 9*CIN (k = c*9 + tap), padded to 2*S; lane (i, kk) feeds k = 2*s + kk of MFMA step s.
//   Packed weights: wpk[nb][s][lane] = W[32*nb + (lane&31)][c][tap] for k = 2*s + (lane>>5) < K, else 0.
//   Workgroup = 4 waves, tile = 8 rows x 32 pixels; wave w owns rows 2w, 2w+1.
// ------------------------------------------------------------------------------------------------
template <int CIN, int NBT>
__global__ void __launch_bounds__(256) conv3x3_first(W2xcConvDesc d, int tiles_x, int ntiles)
{
    constexpr int ROWS = 8, MB = 2, HW = 34, HH = ROWS + 2;
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
        const int gy = clampi(oy0 + py + d.off_y, 0, d.in_h - 1);      // copyMakeBorder REPLICATE
        const int gx = clampi(ox0 + px + d.off_x, 0, d.in_w - 1);
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
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb][s], b[s], acc[mb], 0, 0, 0);
        const int n = nb * 32 + i;
        const float bv = d.bias[n];
#pragma unroll
        for (int mb = 0; mb < MB; mb++) {
            const int y = oy0 + wave * MB + mb;
            if (y < d.out_h) {
                float *orow = d.out + (long long)y * d.out_rs + n;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int x = ox0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                    if (x < d.out_w) orow[(long long)x * COUT] = leaky(acc[mb][r] + bv);
                }
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
template <int CIN, int COUT>
__global__ void __launch_bounds__(256) conv3x3_last(W2xcConvDesc d, int tiles_x, int ntiles)
{
    constexpr int ROWS = 8, HW = 34, HH = ROWS + 2, NPIX = HH * HW;
    constexpr int NBLK = (NPIX + 15) / 16;
    constexpr int N = 9 * COUT, NB16 = (N + 15) / 16;
    constexpr int GS = N | 1;                  // odd LDS row stride
    constexpr int S4 = CIN / 16;
    __shared__ float G[NBLK * 16 * GS];

    const int tile = xcd_remap(blockIdx.x, ntiles);
    const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
    const int oy0 = tile_y * ROWS, ox0 = tile_x * 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kk = lane >> 4, i = lane & 15;

    float b[S4][4][NB16];
#pragma unroll
    for (int s4 = 0; s4 < S4; s4++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int nb = 0; nb < NB16; nb++) b[s4][j][nb] = d.wpk[((s4 * 4 + j) * NB16 + nb) * 64 + lane];

    for (int blk = wave; blk < NBLK; blk += 4) {
        int q = blk * 16 + i;
        q = q < NPIX ? q : NPIX - 1;
        const int py = q / HW, px = q - py * HW;
        const int gy = clampi(oy0 + py + d.off_y, 0, d.in_h - 1);
        const int gx = clampi(ox0 + px + d.off_x, 0, d.in_w - 1);
        const f32x4 *src = reinterpret_cast<const f32x4 *>(d.in + (long long)gy * d.in_rs + (long long)gx * CIN) + kk;
        f32x4 av[S4];
#pragma unroll
        for (int s4 = 0; s4 < S4; s4++) av[s4] = src[s4 * 4];
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
            d.out[(long long)o * d.out_cs + (long long)y * d.out_rs + (long long)x * d.out_ps] = leaky(v + d.bias[o]);
        }
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
    default: return "conv3x3_direct";
    }
}

static int direct_cout_pad(int cout) { return (cout + DIRECT_CG - 1) / DIRECT_CG * DIRECT_CG; }

size_t w2xc_packed_weight_floats(W2xcKernelKind kind, int cin, int cout)
{
    switch (kind) {
    case W2XC_K_MFMA: return (size_t)9 * cin * cout;
    case W2XC_K_FIRST: return (size_t)(cout / 32) * ((9 * cin + 1) / 2) * 64;
    case W2XC_K_LAST: return (size_t)(cin / 16) * 4 * ((9 * cout + 15) / 16) * 64;
    default: return (size_t)cin * 9 * direct_cout_pad(cout);
    }
}

void w2xc_pack_weights(W2xcKernelKind kind, int cin, int cout, const float *w, float *dst)
{
    auto W = [&](int o, int i, int tap) { return w[((size_t)o * cin + i) * 9 + tap]; };
    memset(dst, 0, w2xc_packed_weight_floats(kind, cin, cout) * sizeof(float));
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
static hipError_t launch_mfma(const W2xcConvDesc &d, hipStream_t stream)
{
    constexpr int ROWS = MB * WM;
    const int tiles_x = (d.out_w + 31) / 32, tiles_y = (d.out_h + ROWS - 1) / ROWS;
    const int ntiles = tiles_x * tiles_y;
    const size_t lds_bytes = (size_t)(ROWS + 2) * 34 * 36 * sizeof(float);
    hipLaunchKernelGGL((conv3x3_mfma<CIN, COUT, MB, NB, WM, WN>), dim3(ntiles), dim3(WM * WN * 64), lds_bytes, stream,
                       d, tiles_x, ntiles);
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

hipError_t w2xc_launch_conv(W2xcKernelKind kind, const W2xcConvDesc &d, hipStream_t stream)
{
    if (d.out_w <= 0 || d.out_h <= 0) return hipSuccess;
    if (kind == W2XC_K_MFMA) {
        if (d.in_ps != d.cin || d.in_cs != 1 || d.out_ps != d.cout || d.out_cs != 1) return hipErrorInvalidValue;
        const int key = d.cin * 1000 + d.cout;
        switch (key) {
        //                           CIN  COUT  MB NB WM WN
        case 32032:  return launch_mfma<32, 32, 2, 1, 4, 1>(d, stream);
        case 32064:  return launch_mfma<32, 64, 2, 2, 4, 1>(d, stream);
        case 32128:  return launch_mfma<32, 128, 2, 2, 2, 2>(d, stream);
        case 64032:  return launch_mfma<64, 32, 2, 1, 4, 1>(d, stream);
        case 64064:  return launch_mfma<64, 64, 2, 2, 4, 1>(d, stream);
        case 64128:  return launch_mfma<64, 128, 2, 2, 2, 2>(d, stream);
        case 128032: return launch_mfma<128, 32, 2, 1, 4, 1>(d, stream);
        case 128064: return launch_mfma<128, 64, 2, 2, 4, 1>(d, stream);
        case 128128: return launch_mfma<128, 128, 2, 2, 2, 2>(d, stream);
        default: return hipErrorInvalidValue;
        }
    }
    if (kind == W2XC_K_FIRST) {
        if (d.out_ps != d.cout || d.out_cs != 1) return hipErrorInvalidValue;
        const int key = d.cin * 1000 + d.cout;
        switch (key) {
        case 1032: return launch_tiled8(conv3x3_first<1, 1>, d, stream);
        case 1064: return launch_tiled8(conv3x3_first<1, 2>, d, stream);
        case 1128: return launch_tiled8(conv3x3_first<1, 4>, d, stream);
        case 3032: return launch_tiled8(conv3x3_first<3, 1>, d, stream);
        case 3064: return launch_tiled8(conv3x3_first<3, 2>, d, stream);
        case 3128: return launch_tiled8(conv3x3_first<3, 4>, d, stream);
        default: return hipErrorInvalidValue;
        }
    }
    if (kind == W2XC_K_LAST) {
        if (d.in_ps != d.cin || d.in_cs != 1) return hipErrorInvalidValue;
        const int key = d.cin * 1000 + d.cout;
        switch (key) {
        case 32001:  return launch_tiled8(conv3x3_last<32, 1>, d, stream);
        case 64001:  return launch_tiled8(conv3x3_last<64, 1>, d, stream);
        case 128001: return launch_tiled8(conv3x3_last<128, 1>, d, stream);
        case 32003:  return launch_tiled8(conv3x3_last<32, 3>, d, stream);
        case 64003:  return launch_tiled8(conv3x3_last<64, 3>, d, stream);
        case 128003: return launch_tiled8(conv3x3_last<128, 3>, d, stream);
        default: return hipErrorInvalidValue;
        }
    }
    const int cp = direct_cout_pad(d.cout);
    dim3 grid((d.out_w + 63) / 64, (d.out_h + 3) / 4, cp / DIRECT_CG);
    hipLaunchKernelGGL(conv3x3_direct, grid, dim3(64, 4), 0, stream, d, cp);
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
