// w2xc_first2_wino4.hip -- conv3x3_first2_wino4: layers 1 (ONE plane -> 32) and 2 (32 -> 32) of convertWithModelsBasic's loop
// (/root/reference/src/convertRoutine.cpp:66-76) in ONE launch: layer 1's activations never reach HBM (N3).  Layer 2 -- the 3x3 x 32 x 32
// contraction of Model::filterWorker (/root/reference/src/modelHandler.cpp:117-159) -- runs as Winograd F(4x4, 3x3) on v_mfma_f32_16x16x4_f32 with its
// weights STATIONARY IN REGISTERS; layer 1 (9 multiplies per value) is computed on the fly, per Winograd patch, from a tile of the source plane.
//
//   With 32 input planes the transformed weights of a 32-output-plane layer are 36 x 32 x 32 floats = 144 KiB = 144 registers per lane of a
//   four-wave workgroup: they are loaded ONCE per workgroup and never move again (conv3x3_wino4 streams its U through LDS for every tile: 36 KiB per
//   4-channel stage, its largest cost beside the MFMAs).  What is left per tile is what depends on the pixels.
//
//   Workgroup   4 waves, two workgroups per CU (two waves per SIMD, 256 registers each; 74 KiB of LDS each).  Wave g owns the positions
//               xi in [9 g, 9 g + 9) of the transformed domain (xi = 6 i + j) for all 32 output planes and all 32 channels:
//               U[9][2 plane tiles][8 k-steps] = 144 registers in MFMA A-fragment order (lane = 16 k + plane).
//   Tile        16 blocks of 4x4 = 8 rows x 32 pixels of output (2 x 8 blocks); persistent workgroups walk an XCD-chunked tile list in strips.
//   S           the tile's 12 x 38 source pixels into LDS: replicate padding (copyMakeBorder, convertRoutine.cpp:35,96) and the nearest-2x
//               (main.cpp:132-140, in_shift) folded into the addresses, exactly as conv3x3_first does.
//   T           per lane two patches (block, channel c) and (block, c + 16): the 6 x 6 layer-1 activations leaky(b1[c] + sum w1[c][tap] src) as a
//               bias-first fma chain in tap order over an 8 x 8 window (rolling three rows), then V = B^T d B, written to LDS in B-fragment order
//               V[xi][k-step / 4][lane = 16 k + block][k-step % 4]: wave g handles the channels = g mod 4 with a lane map of its own (conflict-free dword writes).
//   G           36 x (32 planes x 16 blocks x 32 channels) GEMMs: per wave 9 xi x 2 plane tiles x 8 k-steps = 144 MFMAs, one ds_read_b128 per
//               (xi, four k-steps); every V value is read by exactly one wave.
//   X           the accumulators change owner through LDS (over V): M[xi][plane pair][lane = 16 (plane quad) + block] as 8-byte pairs.
//   O           output transform Y = A^T M A, bias, LeakyReLU: two (plane, block) pairs per lane, stores as 16-byte pixel quads of a plane row
//               (eight lanes = one 128-byte line), spread between the arithmetic (a burst of stores costs the wave their issue and the wait for their data registers back to back; the memory pipeline's FIFOs never fill).
//   The phases of a tile run one after the other between workgroup barriers; the second workgroup of the CU is in another phase.  MFMA and VALU
//   share the SIMD's issue time on this hardware (an fp32 MFMA and a VALU instruction of two waves do not run side by side), so the kernel's
//   time is the SUM of its matrix and vector work: what fusion buys is layer 1's HBM round trip (1.07 GB written and read again) and every
//   global load of layer 2 -- measured, the unfused form of this kernel lost as much to its patch loads as it gained (profiles/r5_sweeps.log).
//   Edges       source rows / columns clamped (replicate) in the tile fill; patch columns >= layer 2's input width are zeroed (they only reach
//               outputs >= out_w), rows beyond it are layer-1 values of clamped source rows (they only reach outputs >= out_h).  Blocks sit on
//               rows = 0 mod 4 of layer 2's whole output (W2xcConvDesc::wino_py): banding-invariant on run_rows' four-rows-per-layer geometry.
#include "w2xc_kernels.h"
#include "w2xc_device.h"
#include "w2xc_wino4_math.h"

#include <stdlib.h>
#include <string.h>

#include <atomic>

#ifndef W4S_ABL
#define W4S_ABL 0   // timing-only ablations (wrong results): 1 no input transform arithmetic | 2 no layer-1 arithmetic | 4 no MFMAs | 8 no output transform | 16 no stores
#endif
#ifdef W4S_TIMING
// tools/ubench/first2_wino4_timing.hip: s_memtime stamps of workgroup 0, wave 0: [tile * 8 + k], k = 0 tile start, 1 source tile in LDS, 2 T done, 3 barrier, 4 G done, 5 barrier + X done, 6 barrier, 7 O done
__device__ unsigned long long w4s_stamps[4096];
#define W4S_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && 8 * tn + (k) < 4096) w4s_stamps[8 * tn + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W4S_STAMP(k) do { } while (0)
#endif

// d.in / in_* / in_h / in_w / off_* / in_shift: LAYER 1's input plane (off = layer 1's offsets + layer 2's); d.w1pk / d.bias1: layer 1's W2XC_K_FIRST
// image and bias; d.wpk / d.bias / d.out* / out_h / out_w / wino_py: layer 2
__global__ void __launch_bounds__(256, 2) conv3x3_first2_wino4(W2xcConvDesc d, int tiles_x, int ntiles)
{
    constexpr int STRIP = 16;
    constexpr int SW = 40;                                    // floats per row of the source tile (38 used)
    constexpr unsigned SRC_FLOATS = 36 * 2 * 64 * 4;          // the source tile sits behind the 72 KiB V / M buffer
    constexpr unsigned W1_FLOATS = SRC_FLOATS + 12 * SW;      // layer 1's weights and bias behind it: [channel][9 taps, bias, 2 pad]
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, hi = lane >> 4;
    const int r = n >> 3, c8 = n & 7;                         // block (row, column) of the tile

    // ---- schedule: XCD x gets one contiguous chunk of the tile list; its workgroups take every per-th tile ----
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
    const int cq = ntiles >> 3, cr = ntiles & 7;
    const int chunk_begin = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
    const int chunk_end = chunk_begin + cq + (xcd < cr ? 1 : 0);
    if (chunk_begin + slot >= chunk_end) return;
    const int tiles_y = ntiles / tiles_x;
    auto tile_coords = [&](int t_, int &ty_, int &tx_) {     // strips of 16 tiles, row by row inside a strip
        const int per_strip = STRIP * tiles_y;
        int sidx = t_ / per_strip;
        const int nfull = tiles_x / STRIP;
        if (sidx > nfull) sidx = nfull;
        const int wid = sidx < nfull ? STRIP : tiles_x - nfull * STRIP;
        const int q = t_ - sidx * per_strip;
        ty_ = q / wid;
        tx_ = sidx * STRIP + (q - ty_ * wid);
    };

    // ---- layer 1's weights into LDS (read per patch: 144 registers hold layer 2's) ----
    for (int idx = threadIdx.x; idx < 32 * 12; idx += 256) {
        const int c = idx / 12, e = idx - c * 12;
        lds[W1_FLOATS + idx] = e < 9 ? d.w1pk[(e >> 1) * 64 + (e & 1) * 32 + c] : e == 9 ? d.bias1[c] : 0.0f;   // the W2XC_K_FIRST image: [k-step][lane = 32 (k & 1) + plane]
    }
    // ---- layer 2's weights: U_xi[plane 16 pt + o][channel 4 ks + k] at [wave][xi - 9 wave][pt][ks][lane = 16 k + o], loaded once ----
    float U[9][2][8];
    {
        const float *up = d.wpk + ((size_t)wave * 144) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 144; i++) U[i / 16][(i / 8) & 1][i & 7] = up[(size_t)i * 64];
#pragma unroll
        for (int i = 0; i < 144; i++) asm volatile("" : "+v"(U[i / 16][(i / 8) & 1][i & 7]));   // (register values from here on: never re-loaded)
    }
    // O phase: wave (pt_o, ep) handles the plane pairs o0 = 16 pt_o + 4 hi + 2 ep, o0 + 1 of every block: the wave-uniform part of the plane in a
    // scalar base, the lane's 4 hi planes + row + pixel in a 32-bit byte offset (the launcher checks the range)
    const int pt_o = wave >> 1, ep = wave & 1;
    const int o0 = 16 * pt_o + 4 * hi + 2 * ep;
    float bias0 = d.bias[o0], bias1 = d.bias[o0 + 1];
    // (register VALUES from here on: left pending, the compiler waits for these two loads at every use inside the tile loop -- s_waitcnt vmcnt(0) in front
    //  of each of the O phase's eight stores, i.e. every store waited for the one before it to be acknowledged: round 6, found in the ISA)
    asm volatile("" : "+v"(bias0), "+v"(bias1));
    char *obase[2];
#pragma unroll
    for (int pl = 0; pl < 2; pl++) obase[pl] = reinterpret_cast<char *>(d.out + (long long)(16 * pt_o + 2 * ep + pl) * d.out_cs);
    const unsigned o_hoff = (unsigned)(4 * hi) * (unsigned)d.out_cs * 4u;
    const unsigned out_rs4 = (unsigned)d.out_rs * 4u;
    const int in_w2 = d.out_w + 2;                            // layer 2's input width = layer 1's output width

    int tn = 0;
    (void)tn;
    for (int t = chunk_begin + slot; t < chunk_end; t += per, tn++) {
        int ty, tx;
        tile_coords(t, ty, tx);
        W4S_STAMP(0);
        // ================= S: the 12 x 38 source pixels of the tile (clamped = replicate padding; >> in_shift = nearest 2x) =================
        {
            const int ys = ty * 8 - d.wino_py + d.off_y, xs = tx * 32 + d.off_x;
            // LDS-DMA, one dword per lane, both passes in flight at once and no data register (as a loop of load / s_waitcnt vmcnt(0) / ds_write this phase
            // paid two memory round trips one after the other)
            constexpr int SP = (12 * SW + 255) / 256;
            const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
#pragma unroll
            for (int t = 0; t < SP; t++) {
                const int idx = (int)threadIdx.x + 256 * t;
                if (idx < 12 * SW) {
                    const int py = idx / SW, px = idx - py * SW;
                    const int gy = clampi(ys + py, 0, d.in_h - 1) >> d.in_shift;
                    const int gx = clampi(xs + px, 0, d.in_w - 1) >> d.in_shift;
                    lds_dma4(d.in + ((long long)gy * d.in_rs + gx), lds0 + (SRC_FLOATS + 256 * t + 64 * wave) * 4);
                }
            }
            W2XC_WAIT_VMCNT(0);
        }
        __syncthreads();
        W4S_STAMP(1);
        // ================= T: layer 1 on the fly + input transform -> V: patches (block n, channel wave + 4 hi) and (block n, + 16) =================
        {
            // T's own lane map: block column in bits 0-2, channel step in bits 3-4, block row in bit 5 -- a ds_write_b32 is served in two groups of 32
            // lanes over 32 banks, and lanes 0-31 = (8 block columns x 4 channel steps) write the dwords 4 column + step: 32 banks, no conflict
            // (with the MFMA-side map block = lane & 15 the blocks n and n + 8 of a group share their banks: 30 % of the kernel's LDS cycles)
            const int tc8 = lane & 7, thi = (lane >> 3) & 3, tr = lane >> 5, tnn = 8 * tr + tc8;
            const float *sw = lds + SRC_FLOATS + (4 * tr) * SW + 4 * tc8;   // the lane's 8 x 8 window: rows 4 r + (0..7), columns 4 c8 + (0..7)
            float *vb = lds + ((16 * wave + tnn) * 4 + thi);                // + ((xi * 2 + p) * 64) * 4 floats
            const bool edge = tx * 32 + 34 > in_w2;                         // (wave-uniform: the tile touches the right edge of layer 2's input)
            const int lim = in_w2 - (tx * 32 + 4 * tc8);                    // patch columns >= lim are outside it: zero
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const int c = wave + 4 * (4 * p + thi);
                float w1[9];
                const f32x4 wa = *reinterpret_cast<const f32x4 *>(lds + W1_FLOATS + c * 12), wb = *reinterpret_cast<const f32x4 *>(lds + W1_FLOATS + c * 12 + 4),
                            wc = *reinterpret_cast<const f32x4 *>(lds + W1_FLOATS + c * 12 + 8);
                w1[0] = wa[0]; w1[1] = wa[1]; w1[2] = wa[2]; w1[3] = wa[3]; w1[4] = wb[0]; w1[5] = wb[1]; w1[6] = wb[2]; w1[7] = wb[3]; w1[8] = wc[0];
                const float b1 = wc[1];
                float dd[36];
                float win[3][8];   // three rows of the window, rolling
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(sw + i * SW), b = *reinterpret_cast<const f32x4 *>(sw + i * SW + 4);
                    win[i % 3][0] = a[0]; win[i % 3][1] = a[1]; win[i % 3][2] = a[2]; win[i % 3][3] = a[3];
                    win[i % 3][4] = b[0]; win[i % 3][5] = b[1]; win[i % 3][6] = b[2]; win[i % 3][7] = b[3];
                    if (i >= 2) {   // patch row i - 2 = source rows i - 2, i - 1, i
                        const int pr = i - 2;
#pragma unroll
                        for (int j = 0; j < 6; j++) {
                            if constexpr ((W4S_ABL & 2) != 0) dd[6 * pr + j] = win[i % 3][j] + win[(i + 1) % 3][j + 1] + win[(i + 2) % 3][j + 2];
                            else {
                                float a1 = b1;   // (bias first, then the nine taps in order: conv3x3_first's accumulation)
#pragma unroll
                                for (int rr = 0; rr < 3; rr++)
#pragma unroll
                                    for (int ss = 0; ss < 3; ss++) a1 = __builtin_fmaf(w1[3 * rr + ss], win[(pr + rr) % 3][j + ss], a1);
                                dd[6 * pr + j] = __builtin_fmaxf(a1, 0.1f * a1);   // LeakyReLU (modelHandler.cpp:148-152)
                            }
                        }
                    }
                }
                if (edge) {
#pragma unroll
                    for (int e = 0; e < 36; e++) dd[e] = (e % 6) < lim ? dd[e] : 0.0f;
                }
                if constexpr (!(W4S_ABL & 1)) {
#pragma unroll
                    for (int i = 0; i < 6; i++) bt6(dd[6 * i + 0], dd[6 * i + 1], dd[6 * i + 2], dd[6 * i + 3], dd[6 * i + 4], dd[6 * i + 5]);   // d B
#pragma unroll
                    for (int j = 0; j < 6; j++) bt6(dd[0 + j], dd[6 + j], dd[12 + j], dd[18 + j], dd[24 + j], dd[30 + j]);                         // B^T (.)
                }
#pragma unroll
                for (int xi = 0; xi < 36; xi++) vb[(xi * 2 + p) * 256] = dd[xi];
            }
        }
        W4S_STAMP(2);
        __syncthreads();
        W4S_STAMP(3);
        // ================= G: M_xi = U_xi V_xi for this wave's nine xi =================
        f32x4 acc[9][2];
        {
            const f32x4 *vr = reinterpret_cast<const f32x4 *>(lds) + (size_t)wave * (9 * 2 * 64) + lane;
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int x3 = 0; x3 < 3; x3++) {
                    f32x4 b[3];
#pragma unroll
                    for (int u = 0; u < 3; u++) b[u] = vr[((3 * x3 + u) * 2 + h) * 64];
#pragma unroll
                    for (int q = 0; q < 4; q++)
#pragma unroll
                        for (int u = 0; u < 3; u++)
#pragma unroll
                            for (int pt = 0; pt < 2; pt++) {
                                if constexpr ((W4S_ABL & 4) != 0) {
                                    if (h == 0 && q == 0) acc[3 * x3 + u][pt] = b[u];
                                } else {
                                    const f32x4 c = (h == 0 && q == 0) ? f32x4{0.0f, 0.0f, 0.0f, 0.0f} : acc[3 * x3 + u][pt];
                                    acc[3 * x3 + u][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[3 * x3 + u][pt][4 * h + q], b[u][q], c, 0, 0, 0);
                                }
                            }
                }
        }
        W4S_STAMP(4);
        // ================= X: the accumulators to their output-transform lanes, over the V slices they came from (a wave's xi slices are read and
        //                   written by that wave alone: no barrier in front).  C/D of the 16x16 MFMA: lane = 16 k' + block, register e = plane
        //                   4 k' + e of the plane tile -> M[xi][pt][e >> 1][lane] as pairs =================
        {
            f32x2 *mw = reinterpret_cast<f32x2 *>(lds) + (size_t)wave * (9 * 4 * 64) + lane;
#pragma unroll
            for (int xl = 0; xl < 9; xl++)
#pragma unroll
                for (int pt = 0; pt < 2; pt++) {
                    mw[((xl * 2 + pt) * 2 + 0) * 64] = f32x2{acc[xl][pt][0], acc[xl][pt][1]};
                    mw[((xl * 2 + pt) * 2 + 1) * 64] = f32x2{acc[xl][pt][2], acc[xl][pt][3]};
                }
        }
        W4S_STAMP(5);
        __syncthreads();
        W4S_STAMP(6);
        // ================= O: Y = A^T M A, bias, LeakyReLU, stores: planes o0, o0 + 1 of block n =================
        {
            const float *mr = lds + ((size_t)((pt_o * 2 + ep) * 64 + lane)) * 2;   // + xi * 512 floats (+ 1: the second plane)
            const int oy = ty * 8 - d.wino_py + 4 * r, ox = tx * 32 + 4 * c8;
            const unsigned so = o_hoff + 4u * (unsigned)ox;
            static_for<0, 2>([&](auto PL) {
                constexpr int pl = decltype(PL)::value;
                float m[36];
#pragma unroll
                for (int xi = 0; xi < 36; xi++) m[xi] = mr[xi * 512 + pl];
                float tt[4][6];
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    if constexpr ((W4S_ABL & 8) != 0) { tt[0][j] = m[j]; tt[1][j] = m[6 + j]; tt[2][j] = m[12 + j]; tt[3][j] = m[18 + j] + m[24 + j] + m[30 + j]; }
                    else at6(m[j], m[6 + j], m[12 + j], m[18 + j], m[24 + j], m[30 + j], tt[0][j], tt[1][j], tt[2][j], tt[3][j]);
                }
                const float bq = pl ? bias1 : bias0;
                static_for<0, 4>([&](auto I_) {
                    constexpr int i = decltype(I_)::value;
                    f32x4 yr;
                    if constexpr ((W4S_ABL & 8) != 0) {
                        yr = f32x4{tt[i][0] + tt[i][4], tt[i][1] + tt[i][5], tt[i][2], tt[i][3]};
                    } else {
                        float y0, y1, y2, y3;
                        at6(tt[i][0], tt[i][1], tt[i][2], tt[i][3], tt[i][4], tt[i][5], y0, y1, y2, y3);
                        const float w0 = y0 + bq, w1 = y1 + bq, w2 = y2 + bq, w3 = y3 + bq;
                        yr = f32x4{__builtin_amdgcn_fmed3f(w0, 0.1f * w0, 3.402823466e+38f), __builtin_amdgcn_fmed3f(w1, 0.1f * w1, 3.402823466e+38f),
                                   __builtin_amdgcn_fmed3f(w2, 0.1f * w2, 3.402823466e+38f), __builtin_amdgcn_fmed3f(w3, 0.1f * w3, 3.402823466e+38f)};
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // whole quads: the row stride holds roundup4(out_w) pixels (the launcher checks), columns >= out_w are never read as data
                    if constexpr (!(W4S_ABL & 16)) {
                        if (oy + i >= 0 && oy + i < d.out_h && ox < d.out_w)
                            *reinterpret_cast<f32x4 *>(obase[pl] + (size_t)(so + (unsigned)(oy + i) * out_rs4)) = yr;
                    } else if (yr[0] == 12345.678f) d.out[lane] = yr[1] + yr[2] + yr[3];
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        }
        W4S_STAMP(7);
        __syncthreads();   // (the next tile's V lands on M, its source tile on this one's)
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
bool w2xc_first2_wino4_supported(int cin1, int cout1, int cout2) { return cin1 == 1 && cout1 == 32 && cout2 == 32; }

// layer 2's weights: wpk[wave g][xi - 9 g][plane tile pt][k-step ks][lane = 16 k + o] = U_xi[plane 16 pt + o][channel 4 ks + k], xi = 6 i + j,
// U = G g G^T formed in double and rounded once.  w is [32][32][3][3] (modelHandler.cpp:102).  36 * 32 * 32 floats.
void w2xc_first2_wino4_pack(const float *w, float *dst)
{
    const int cin = 32, cout = 32;
    for (int plane = 0; plane < cout; plane++)
        for (int c = 0; c < cin; c++) {
            const float *g = w + ((size_t)plane * cin + c) * 9;
            double tmp[6][3];
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 3; j++) tmp[i][j] = W2XC_WINO4_G[i][0] * g[0 * 3 + j] + W2XC_WINO4_G[i][1] * g[1 * 3 + j] + W2XC_WINO4_G[i][2] * g[2 * 3 + j];
            const int pt = plane / 16, o = plane % 16, ks = c / 4, k = c % 4;
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 6; j++) {
                    const double u = tmp[i][0] * W2XC_WINO4_G[j][0] + tmp[i][1] * W2XC_WINO4_G[j][1] + tmp[i][2] * W2XC_WINO4_G[j][2];
                    const int xi = 6 * i + j, gw = xi / 9, xl = xi % 9;
                    dst[((((size_t)gw * 9 + xl) * 2 + pt) * 8 + ks) * 64 + k * 16 + o] = (float)u;
                }
        }
}

// d.in .. in_shift: layer 1's one-plane input (any row stride; in_h / in_w in UPSCALED coordinates when in_shift = 1), off_y / off_x = layer 1's offsets
// plus layer 2's; d.w1pk / d.bias1 = layer 1's W2XC_K_FIRST image / bias; d.wpk = w2xc_first2_wino4_pack image, d.bias, planar fp32 out (out_ps = 1,
// 16-byte aligned rows of >= roundup4(out_w) floats), d.out_h / out_w / wino_py = layer 2's region
hipError_t w2xc_launch_first2_wino4(const W2xcConvDesc &d, hipStream_t stream)
{
    if (d.out_w <= 0 || d.out_h <= 0) return hipSuccess;
    if (d.cin != 32 || d.cout != 32 || !d.w1pk || !d.bias1 || d.in_ps != 1 || d.in_shift < 0 || d.in_shift > 1 || d.out_terms != 0) return hipErrorInvalidValue;
    if (d.out_ps != 1 || (d.out_rs & 3) != 0 || (d.out_cs & 3) != 0 || (((size_t)d.out) & 15) != 0 || d.out_rs < ((d.out_w + 3) & ~3)) return hipErrorInvalidValue;
    // 32-bit lane offsets: 12 plane strides + a row of THIS launch's region + a pixel -- the largest one a lane forms, whatever the relation of the strides
    if (12ll * d.out_cs * 4 + ((long long)d.out_h + 8) * d.out_rs * 4 >= (1ll << 32)) return hipErrorInvalidValue;
    const int tiles_x = (d.out_w + 31) / 32, tiles_y = (d.out_h + (d.wino_py & 3) + 7) / 8;
    const int ntiles = tiles_x * tiles_y;
    constexpr size_t lds_bytes = 36 * 2 * 64 * 16 + 12 * 40 * 4 + 32 * 12 * 4;   // V (then M over it): 72 KiB; the source tile; layer 1's weights
    auto kern = conv3x3_first2_wino4;
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
    if (grid > ((ntiles + 7) & ~7)) grid = (ntiles + 7) & ~7;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, stream, d, tiles_x, ntiles);
    return hipGetLastError();
}
