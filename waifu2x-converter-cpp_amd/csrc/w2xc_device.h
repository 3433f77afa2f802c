// w2xc_device.h -- device-side helpers shared by the gfx950 kernel files (w2xc_kernels.hip, w2xc_split.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// compile-time loop: f(std::integral_constant<int, i>) for i in [B, E).  Register arrays indexed through a lambda PARAMETER
// end up in scratch memory even under #pragma unroll; indices that arrive as integral constants never do.
template <int B, int E, class F>
static __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

static __device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// cv::max / cv::min / cv::scaleAdd(neg, 0.1, pos), modelHandler.cpp:148-152
static __device__ __forceinline__ float leaky(float v) { return v > 0.0f ? v : 0.1f * v; }

// bf16 storage (round-to-nearest-even; finite values only on this path)
typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ bf16_t f2bf(float f)
{
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
static __device__ __forceinline__ void store_act(float *p, float v) { *p = v; }
static __device__ __forceinline__ void store_act(bf16_t *p, float v) { *p = f2bf(v); }

// XCD-aware block remap: the dispatcher places block b on XCD b % 8; give each XCD one contiguous
// range of tiles so neighbouring tiles (which share halo rows/columns) share an L2.  Bijective
// for any grid size (cdna_hip_programming.md T1).
static __device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// ---- LDS-DMA (global_load_lds_dwordx4) and counted waits ----------------------------------------
// ("VALU writes an SGPR -> VMEM reads it" needs 5 wait states, and the hazard recogniser does not look into an asm statement: where the compiler had just
//  restored a scalar base from a spill lane with v_readlane, the kernels of round 5 read it 4 states later -- tools/check_sgpr_vmem_hazard.py.  The transfers
//  therefore read a COPY of the base made by the scalar ALU inside the statement: VALU -> SALU is interlocked by the hardware, SALU -> VMEM has no hazard, and
//  the s_mov_b64 doubles as the one wait state the write of m0 needs -- no s_nop at all.  Round 6's first fix, s_nop 3 in every helper, cost every
//  conv3x3_wino4 layer 1-1.5 %: r6_sweeps.log 6.)
// m0 carries the wave-uniform LDS byte address of the transfer; it is compiler-reserved and this
// kernel uses it for nothing else, so it is simply overwritten (clobber listed: hipcc only warns).
static __device__ __forceinline__ void lds_dma16(const void *gptr, unsigned lds_byte_addr)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :
                 : "v"(gptr), "s"(lds_byte_addr)
                 : "memory", "m0");
}
// one dword per lane (lane l lands at lds_byte_addr + 4 l): a gather of single pixels straight into LDS, no data register
static __device__ __forceinline__ void lds_dma4(const void *gptr, unsigned lds_byte_addr)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off"
                 :
                 : "v"(gptr), "s"(lds_byte_addr)
                 : "memory", "m0");
}
// scalar base + 32-bit per-lane byte offset + immediate: no 64-bit VALU address per transfer
template <int IMM>
static __device__ __forceinline__ void lds_dma16_s(const void *sbase, unsigned voff, unsigned lds_byte_addr)
{
    unsigned long long b;
    asm volatile("s_mov_b32 m0, %3\n\ts_mov_b64 %0, %2\n\tglobal_load_lds_dwordx4 %1, %0 offset:%4"
                 : "=&s"(b)
                 : "v"(voff), "s"(sbase), "s"(lds_byte_addr), "n"(IMM)
                 : "memory", "m0");
}
// the same with the LDS address as (wave-uniform base + compile-time offset): no SGPR per destination for the compiler to hoist and keep alive
template <unsigned LDS_IMM>
static __device__ __forceinline__ void lds_dma16_si(const void *sbase, unsigned voff, unsigned lds_base)
{
    unsigned long long b;
    asm volatile("s_add_u32 m0, %3, %4\n\ts_mov_b64 %0, %2\n\tglobal_load_lds_dwordx4 %1, %0"
                 : "=&s"(b)
                 : "v"(voff), "s"(sbase), "s"(lds_base), "n"(LDS_IMM)
                 : "memory", "m0", "scc");
}
// ... and an immediate byte offset (-4096 .. 4095): several transfers off ONE scalar base.  The hardware adds the instruction's offset to the LDS address
// as well (LDS address = m0 + offset + lane * 16 -- measured: with m0 = the destination every piece landed GOFF bytes off), so m0 = destination - GOFF
template <unsigned LDS_IMM, int GOFF>
static __device__ __forceinline__ void lds_dma16_sio(const void *sbase, unsigned voff, unsigned lds_base)
{
    static_assert(GOFF >= -4096 && GOFF <= 4095, "global_load_lds immediate offset");
    unsigned long long b;
    asm volatile("s_add_u32 m0, %3, %4\n\ts_mov_b64 %0, %2\n\tglobal_load_lds_dwordx4 %1, %0 offset:%5"
                 : "=&s"(b)
                 : "v"(voff), "s"(sbase), "s"(lds_base), "n"((int)LDS_IMM - GOFF), "n"(GOFF)
                 : "memory", "m0", "scc");
}
// 16-byte store at (scalar base + 32-bit per-lane byte offset): ONE address register however many stores share the lane offset
static __device__ __forceinline__ void store16_s(const void *sbase, unsigned voff, f32x4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, %2" : : "v"(voff), "v"(v), "s"(sbase) : "memory");
}
#define W2XC_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
// s_waitcnt vmcnt(n) for an n that constant-folds after unrolling; the queue holds at most 63 entries
static __device__ __forceinline__ void wait_vmcnt_n(int n)
{
#define W2XC_WC(k) case k: W2XC_WAIT_VMCNT(k); break;
#define W2XC_WC8(k) W2XC_WC(k) W2XC_WC(k + 1) W2XC_WC(k + 2) W2XC_WC(k + 3) W2XC_WC(k + 4) W2XC_WC(k + 5) W2XC_WC(k + 6) W2XC_WC(k + 7)
    switch (n) {
        W2XC_WC8(0) W2XC_WC8(8) W2XC_WC8(16) W2XC_WC8(24) W2XC_WC8(32) W2XC_WC8(40) W2XC_WC8(48)
        W2XC_WC(56) W2XC_WC(57) W2XC_WC(58) W2XC_WC(59) W2XC_WC(60) W2XC_WC(61) W2XC_WC(62)
    default: break;
    }
#undef W2XC_WC8
#undef W2XC_WC
}
