// trsim_pilot.hip — cnn_2d_speed_control in the loop (BASELINE config 5).
//
// Model: Keras_2D_CNN.get_model(input_shape, num_outputs=2) of the reference
// (TritonRacerSim/components/keras_train.py:127-174, chosen for cnn_2d_speed_control at :393-395): conv 5x5/2 x3
// (24, 32, 64), conv 3x3/1 x4 (64, 64, 128, 128), all 'valid' + ReLU, flatten (NHWC order), dense 100/50/25 + ReLU,
// linear output 2.  Dropout is identity at inference.  57.8 M MAC per 120x160 frame.
//
// Convolutions are implicit GEMMs on the matrix cores, one kernel for every layer:
//   C[pixel][cout] = sum_k A[pixel][k] * B[k][cout],  k = (kh, kw, cin) cut into granules of 8 input channels
//   v_mfma_f32_32x32x16_f16: a wave owns 32 output pixels x all (padded) output channels; per 16-deep k-step the
//   lane (row r = lane & 31, half h = lane >> 5) needs A[r][8h .. 8h+7] = ONE granule = one 16-byte load straight
//   from the NHWC fp16 activation in global memory (no im2col buffer, no LDS for A); B granules [g][cout][8] are
//   staged per workgroup in LDS in exactly the fragment order, so a B fragment is one ds_read_b128.
//   conv1 reads the uint8 frame directly: a k-step is one kernel row = 15 contiguous bytes (5 px x RGB) + 1 pad,
//   fetched with 3 aligned dwords + v_alignbyte, converted exactly to binary16; the 1/255 of the pilot's normalisation
//   (components/keras_pilot.py:49-50) is folded into conv1's weights as 256/255, with 2^-8 in its epilogue (kConv1Scale).
//   dense1 (4608 -> 100) is the same kernel as a 1x1 convolution over "pixels" = frames.
// The fp32 tail (dense2, dense3, output) and KerasPilot's post-processing (keras_pilot.py:78-95; calcThrottle /
// calcBreak of utils/mapping.py:23-35) run in one small kernel that writes the env's next controls.
//
// Numerics: binary16 (fp16) operands, fp32 accumulate, activations stored as binary16 (saturating at 65504) — checked against a
// PyTorch fp32 reference with the same fp16-rounded weights (tests/test_pilot.py, tolerance stated there).  Until round 2 the
// 16-bit format was bfloat16: the same MFMA rate, but 8 significant bits instead of 11 — measured max |output - fp32| 5.0e-4 against
// 4.4e-5 (profiles/r03_pilot_precision.txt); the reference's arithmetic is fp32 (keras_pilot.py:49-55).  Bound: MFMA.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <memory>
#include <vector>

#include "../../include/trsim.h"
#include "trsim_internal.hpp"

#define TRS_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;   // 8 binary16 values: one MFMA operand fragment per lane (round 3: binary16 instead of bfloat16 —
                                                              // the same MFMA rate, 3 more mantissa bits: max |output - fp32| 5e-4 -> 4e-5, profiles/r03_pilot_precision.txt)
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned u4v __attribute__((ext_vector_type(4)));

constexpr int kLayers = 11;               // conv1..7, dense1..3, output
#ifndef TRS_CONV_ABLATE
#define TRS_CONV_ABLATE 0   /* diagnostic builds of trs_conv_lt_kernel, never shipped: 1 = no steady-state global loads, 2 = no LDS transpose, 3 = no MFMA, 4 = no stores */
#endif
constexpr int kConvPrefetch = 4;           // trips (pairs of k-steps) of pixel fragments in flight per wave

struct ConvParams {
    const void* in;            // fp16 NHWC [N][IH][IW][CIN]   (conv1: uint8 [N][IH][IW][3])
    const u4v* w;              // [G_pad][COUT_PAD] granules of 8 fp16
    const float* bias;         // [COUT_PAD]
    const int* goff;           // [G_pad] byte offset of granule g relative to the output pixel's input base
    void* out;                 // fp16 NHWC [N][OH][OW][COUT]  or float when out_f32
    int in_bytes;              // size of `in` (buffer-load bounds)
    int N, IH, IW, CIN, OH, OW, COUT, COUT_PAD, S;
    int G, G_pad, M;           // granules (even-padded), GEMM rows = N*OH*OW
    int gchunk;                // granules per LDS stage (even)
    int relu, out_f32, in_px_bytes;   // in_px_bytes: bytes per input pixel (CIN*2, conv1: 3)
    int ksplit;                // > 1: blockIdx.y owns a slice of the granules and writes its partial sums to slab blockIdx.y of `out` (fp32 [ksplit][M][COUT], no ReLU)
    int nt_out;                // non-temporal activation stores (outputs far larger than L2; measured: conv1 86 -> 80 us, small layers lose)
    int KH, KW, run_pad, cg, span_nl;   // span kernel: kernel rows, granules per kernel row (padded), granules per pixel, load instructions per kernel row
    float oscale;              // the sums are multiplied by this power of two inside the bias FMA: 2^-8 for conv1 (kConv1Scale), 1 elsewhere
};

extern __shared__ __attribute__((aligned(16))) unsigned char psmem[];
typedef const __attribute__((address_space(3))) u4v* lds_u4vp;   // an LDS address held as an integer: no "+ psmem" per access

__device__ __forceinline__ unsigned short f2h(float f)
{   // round to nearest even; beyond binary16's range: 65504 (the activations saturate, they never become infinite)
    const _Float16 h = (_Float16)__builtin_fminf(__builtin_fmaxf(f, -65504.0f), 65504.0f);
    return __builtin_bit_cast(unsigned short, h);
}

typedef __attribute__((ext_vector_type(2))) _Float16 h16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ unsigned pack_h16x2(float lo, float hi)
{   // round to nearest even, two values per dword: v_cvt_pk_f16_f32 on gfx950 (beyond 65504: +-infinity; the ReLU form below saturates)
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h16x2));
}

// two uint8 values (already binary32, 0..255: exact in binary16 under any rounding) as a binary16 pair: v_cvt_pkrtz_f16_f32
__device__ __forceinline__ unsigned u8pair_h16(float f0, float f1) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(f0, f1)); }

constexpr float kConv1Scale = 1.0f / 256.0f;   // conv1's weights carry 256 / 255 (the pilot's / 255 of keras_pilot.py:49-50 x 2^8, so that small weights stay normal
                                               // binary16 numbers); its epilogues multiply the sum by 2^-8 inside the bias FMA: exact, no extra instruction

typedef short s16x2 __attribute__((ext_vector_type(2)));
// ReLU on two packed binary16: the sign bit is the int16's, so max(., 0) as int16 zeroes the negatives (and -0).  One v_pk_max_i16
// for two values instead of a v_max_f32 each in front of the conversion; round-to-nearest keeps the sign, so the bits are the same.
// Non-negative binary16 values order like their int16 patterns, so min(., 0x7BFF) turns an overflowed +infinity (0x7C00) into 65504:
// activations saturate instead of poisoning the next layer (one more v_pk_min_i16 per pair).
// (The epilogues are vector-ALU work between short MFMA runs: conv1's tile of 5 MFMAs carried ~65 VALU instructions.)
__device__ __forceinline__ unsigned relu_h16x2(unsigned packed)
{
    const s16x2 v = __builtin_bit_cast(s16x2, packed), z = {0, 0}, top = {0x7BFF, 0x7BFF};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_elementwise_max(v, z), top));
}
__device__ __forceinline__ uint2 relu_pack4(float v0, float v1, float v2, float v3)
{
    return make_uint2(relu_h16x2(pack_h16x2(v0, v1)), relu_h16x2(pack_h16x2(v2, v3)));
}
// ... without the saturation, for sums that provably stay inside binary16 (conv1 of the fused head: |bias| + sum |w| < 65504 is checked when the weights
// are loaded — its inputs are pixels / 256 <= 1)
__device__ __forceinline__ uint2 relu_pack4_bounded(float v0, float v1, float v2, float v3)
{
    const s16x2 z = {0, 0};
    return make_uint2(__builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pack_h16x2(v0, v1)), z)),
                      __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pack_h16x2(v2, v3)), z)));
}

// The epilogues' half exchange.  Register quad q of a 32x32 accumulator block holds channels 8q + 4h .. + 3 of pixel r in lane (r, h): the two lanes of a
// pixel hold the two halves of every 8-channel group.  So that each lane stores whole 16-byte groups — lane h = 0 groups 0 and 1, lane h = 1 groups 2
// and 3 — lane h = 0 needs its partner's half of groups 0, 1 and lane h = 1 its partner's half of groups 2, 3.  v_permlane32_swap_b32 a, b exchanges
// the upper 32 lanes of a with the lower 32 lanes of b (gfx950): with a = a word of group q and b = the same word of group q + 2, BOTH halves of the wave
// end with (own half, partner's half) of the group they store, in (a, b) order — one full-rate VALU instruction per word where rounds 2-3 used a select,
// a ds_bpermute through the LDS crossbar and two more selects (round 4; the same bytes in the same places).
__device__ __forceinline__ void half_swap(unsigned& a, unsigned& b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
}
__device__ __forceinline__ void quad_groups(const uint2 (&w)[4], u4v& g0, u4v& g1)
{
    unsigned a0 = w[0].x, a1 = w[0].y, b0 = w[2].x, b1 = w[2].y, c0 = w[1].x, c1 = w[1].y, d0 = w[3].x, d1 = w[3].y;
    half_swap(a0, b0); half_swap(a1, b1); half_swap(c0, d0); half_swap(c1, d1);
    g0 = u4v{a0, a1, b0, b1};                               // h = 0: channels 0..7 of group 0; h = 1: channels 0..7 of group 2
    g1 = u4v{c0, c1, d0, d1};                               // h = 0: group 1; h = 1: group 3
}
// The accumulators of a wave item start at the layer's bias (the MFMA's C operand adds it) instead of at zero with 16 v_add per 32 x 32 block in
// the epilogue: register 4 q + j of block nb = channel cbase + nb*32 + 8 q + 4 h + j.  (fp32 sum order: bias first — the same fp16 activations up to
// the last bit of the fp32 sum; tests/test_pilot.py compares against the PyTorch mirror.)
template <int NT, int NB>
__device__ __forceinline__ void acc_from_bias(f32x16 (&acc)[NT][NB], const float4* lbias, int cbase, int h)
{
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b = lbias[(cbase + nb * 32 + 8 * q + 4 * h) >> 2];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { acc[nt][nb][4 * q] = b.x; acc[nt][nb][4 * q + 1] = b.y; acc[nt][nb][4 * q + 2] = b.z; acc[nt][nb][4 * q + 3] = b.w; }
        }
}

#include "trsim_pilot_layers.hpp"   // the single-layer kernels: conv1 as its own layer, the span kernel (conv2 unfused, conv3 at 240x320), the quad-load fallback

struct FrameConvParams {
    const u4v* in;             // fp16 NHWC [N][IH][IW][CIN] as 16-B granules
    const u4v* w;              // [KH*KW*cg][COUT_PAD] granules (kernel-row major, then tap, then channel granule)
    const float* bias;
    unsigned short* out;       // fp16 NHWC [N][OH][OW][COUT]
    int N, IH, IW, OH, OW, COUT, COUT_PAD, KH, KW;
    int F, cg, cgs;            // units (frames, or row bands of frames) per workgroup; granules per pixel (8 / 16) and log2 of it
    int bands, ohb, ihb;       // a frame whose activation does not fit LDS is cut into `bands` bands of ohb output rows = ohb + KH - 1 input rows
    int relu;                  // (neighbouring bands re-stage KH - 1 rows); bands == 1: ohb = OH, ihb = IH
    unsigned magic_uout, magic_ow, magic_bands;   // floor(p / d) = umulhi(p, ceil(2^32 / d)), exact for p < 2^32 / d: d = ohb x OW, OW, bands (an item's set-up without divisions)
};
// weight prefetch: a register ring of R k-steps (R x NB granules per lane), refilled slot by slot right behind the MFMAs that
// consumed the slot: R x NT x NB x 32 MFMA clocks of lead time against an L2 round trip of 500-800 (R = 4: one tap at 64 input
// channels, half a tap at 128; rings of 6 and 8 measured no faster).

#ifndef TRS_FRAME_STAMPS
#define TRS_FRAME_STAMPS 0   /* diagnostic build, never shipped: workgroup 7 prints the s_memtime ticks of its staging, work and waits */
#endif
#ifndef TRS_FRAME_LOADERS
#define TRS_FRAME_LOADERS 4   /* loader waves of trs_conv_frame_kernel beside its 8 compute waves */
#endif
#ifndef TRS_FRAME_R
#define TRS_FRAME_R 4   /* weight ring depth of trs_conv_frame_kernel in k-steps */
#endif
#ifndef TRS_FRAME_ABLATE
#define TRS_FRAME_ABLATE 0   /* timing-only diagnostic builds of trs_conv_frame_kernel, never shipped: 1 = no weight refills, 3 = no stores, 4 = no staging */
#endif
__device__ __forceinline__ int frame_swz(int pix, int cgs) { return cgs == 3 ? ((pix >> 1) & 7) : (pix & 15); }

// Round 4: persistent and double-buffered, like trs_conv_frame5_kernel.  Until then a workgroup staged its F units, computed them and ended; its
// ~150-200 registers allow one 8-wave workgroup per CU, so nothing ran beside the staging: a timing-only build without it (-DTRS_FRAME_ABLATE=4)
// put it at 16-25 % of the 240x320 layers (conv4 49.1 -> 36.8 us, conv5 39.2 -> 29.8, conv6 58.9 -> 49.6, conv7 99.4 -> 83.7 per 512 frames).
// Now one workgroup per CU walks its groups of F units: COMPUTE waves work on group i from one LDS buffer while LOADERS waves run the LDS-DMA
// of group i + 1 into the other; one barrier per group.  The weights come by buffer load (one lane offset + a scalar per k-step: no 64-bit
// address per ring slot — ~60 registers less, which is what lets 12 waves of <= 168 registers share the CU).
template <int NT, int NB, int HALF, int R, int COMPUTE, int LOADERS>
__global__ __launch_bounds__(64 * (COMPUTE + LOADERS), 1) void trs_conv_frame_kernel(const FrameConvParams p)
{
    static_assert(R >= 1 && R <= 9 * HALF, "ring depth in k-steps");
    constexpr int nwaves = COMPUTE + LOADERS, BLOCK = 64 * nwaves;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int n_units = p.N * p.bands, n_groups = (n_units + p.F - 1) / p.F;
    const int upix = p.ihb * p.IW, uout = p.ohb * p.OW;                     // input pixels staged / output pixel slots per unit
    const unsigned buf_bytes = (unsigned)((p.F * upix) << p.cgs) * 16u;     // one group of units in LDS
    const unsigned lds_base = (unsigned)(uintptr_t)psmem;
    // staging of group g into the buffer at byte offset `off` by waves w0 .. w0 + nw - 1 (LDS-DMA: 1 KB per wave instruction; the XOR swizzle goes
    // on the SOURCE address, the LDS side stays lane-linear)
    // A unit's input rows are whole rows of one frame: ONE contiguous run of upix x cg granules in memory (the rows a last band asks for below its
    // frame are never read by a valid pixel: their addresses are clamped into the activation).  So a slot's source is base + slot ^ swizzle — a shift,
    // a mask and an XOR per DMA instruction (the general (unit, row, column) split of rounds 2-3 cost ~80 instructions per 1 KB: four loader waves
    // then needed as long for a group as eight compute waves for its MFMAs).
    const int last_gran = (int)(((size_t)p.N * p.IH * p.IW) << p.cgs) - 1;
    auto stage = [&](int g, unsigned off, int w0, int nw) {
        const int u0 = g * p.F, nu = min(p.F, n_units - u0);
        const int total = upix << p.cgs;                                    // granule slots of one unit
        for (int ul = 0; ul < nu; ++ul) {
            const int u = u0 + ul, f = u / p.bands, band = u - f * p.bands;  // (wave-uniform: scalar unit)
            const int base = ((f * p.IH + band * p.ohb) * p.IW) << p.cgs;     // first granule of the unit in the activation (< 2^31: in_bytes fits an int)
            const unsigned dst = lds_base + off + (unsigned)(ul * total) * 16u;
            for (int s0 = (wave - w0) * 64; s0 < total; s0 += nw * 64) {
                const int sl = s0 + lane;
                if (sl < total && TRS_FRAME_ABLATE != 4) {
                    const int pix = ul * upix + (sl >> p.cgs), q = sl & (p.cg - 1);   // (the swizzle follows the pixel's index in the GROUP, as the readers compute it)
                    const int src = min(base + (sl - q) + (q ^ frame_swz(pix, p.cgs)), last_gran);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.in + src),
                                                     (__attribute__((address_space(3))) void*)(uintptr_t)(dst + s0 * 16), 16, 0, 0);
                }
            }
        }
    };
    float* lb = reinterpret_cast<float*>(psmem + 2 * (size_t)buf_bytes);    // the bias behind the two buffers
    const float4* lbias = reinterpret_cast<const float4*>(lb);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<u4v*>(p.w), 0, 9 * p.cg * p.COUT_PAD * 16, 0x00020000);   // [9 taps x cg granules][COUT_PAD] granules
#if TRS_FRAME_STAMPS
    long long fs_t0 = (long long)__builtin_amdgcn_s_memtime(), fs_st = 0, fs_k = 0, fs_w = 0;
#endif
    int g = blockIdx.x;
    if (g < n_groups) stage(g, 0u, 0, nwaves);                              // the first group: all waves
    for (int i = tid; i < p.COUT_PAD; i += BLOCK) lb[i] = p.bias[i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#if TRS_FRAME_STAMPS
    { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); fs_st += t_ - fs_t0; fs_t0 = t_; }
#endif
    constexpr int ksteps = 9 * HALF;                                        // k-steps (16 input channels of one tap each) of a 3x3 kernel
    const int n_cgrp = p.COUT_PAD / (NB * 32);
    for (int it = 0; g < n_groups; ++it, g += gridDim.x) {
        const unsigned cur = (it & 1) ? buf_bytes : 0u;
        if (wave >= COMPUTE) {                                              // loaders: the next group into the other buffer
            if (g + (int)gridDim.x < n_groups) stage(g + (int)gridDim.x, buf_bytes - cur, COMPUTE, LOADERS);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            const int u0 = g * p.F, nu = min(p.F, n_units - u0);
            int m_wg = nu * uout;                                           // output pixel slots of this group
            if (p.F == 1 && p.bands > 1) {                                  // one unit: a short last band has no tiles for the rows below the frame
                const int band = u0 % p.bands;
                m_wg = min(p.ohb, p.OH - band * p.ohb) * p.OW;
            }
            const int n_tiles = (m_wg + NT * 32 - 1) / (NT * 32);
            for (int item = wave; item < n_tiles * n_cgrp; item += COMPUTE) {
                const int cgrp = item / n_tiles, tile = item - cgrp * n_tiles;
                const int cbase = cgrp * NB * 32;
                int lbase[NT];                                              // linear LDS pixel index of each of this lane's windows
                long long obase[NT];                                        // output pixel index in the layer's activation, -1 = no such pixel
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int m = tile * NT * 32 + nt * 32 + r, mc = min(m, m_wg - 1);
                    const int ul = (int)__umulhi((unsigned)mc, p.magic_uout), rem = mc - ul * uout, oyl = (int)__umulhi((unsigned)rem, p.magic_ow), ox = rem - oyl * p.OW;
                    const int u = u0 + ul, f = p.magic_bands ? (int)__umulhi((unsigned)u, p.magic_bands) : u, oy = (u - f * p.bands) * p.ohb + oyl;   // (magic 0: one band per frame)
                    lbase[nt] = (ul * p.ihb + oyl) * p.IW + ox;
                    obase[nt] = (m < m_wg && oy < p.OH) ? ((long long)f * p.OH + oy) * p.OW + ox : -1;
                }
                // weight granule (2 k + h, cout cbase + nb*32 + r) by buffer load: one lane offset + a scalar per (k-step, block)
                // (the k-step part of the offset is ONE scalar that runs with the loads — a constant per (k-step, block), 144 of them at 128 input channels, was hoisted
                // out of the item loop and spilled to vector-register lanes: a v_readlane + wait states in front of every load; the block is the immediate offset)
                const int wvoff = (h * p.COUT_PAD + cbase + r) * 16;
                int wso = 0;
                auto wload = [&](int nb) { return __builtin_bit_cast(u4v, __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff + nb * 512, wso, 0)); };
                auto wnext = [&]() { wso += 2 * p.COUT_PAD * 16; asm volatile("" : "+s"(wso)); };   // the loads are issued in k order
                u4v ring[R][NB];
#pragma unroll
                for (int d = 0; d < R; ++d) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) ring[d][nb] = wload(nb);
                    wnext();
                }
                f32x16 acc[NT][NB];
                acc_from_bias<NT, NB>(acc, lbias, cbase, h);
                // All ksteps k-steps are unrolled (3 x 12 or 9 x 8 ...): ONE basic block, no back edge — the compiler counts the weight
                // loads in flight exactly (vmcnt(N) per k-step) instead of draining the queue at a loop head or behind a branch.  The pixel
                // fragments are software-pipelined by hand: k-step k + 1's ds_reads are issued BEFORE k-step k's MFMAs (the sched_barrier
                // that keeps "MFMAs of k, then the refill of k's ring slot" in place would otherwise also pin each k-step's LDS reads
                // right in front of its own MFMAs: one LDS latency per 4 MFMAs).
                // (Round 4: a second sched_barrier between those reads and the MFMAs — hipcc sinks the reads behind three of a k-step's four MFMAs — was
                // measured in the frame kernels, the chain and frame5: all slower, chain 46.2 -> 48.4 us, conv7 96.1 -> 100.5: the address arithmetic then runs
                // as a burst with no MFMA beside it.)
                // the XOR-swizzled LDS address of a tap's fragment once per tap: granule 2 j + h of k-step j, and (2 j + h) ^ swizzle = (h ^ swizzle) ^ 2 j — the
                // tap's byte address ^ 32 j, one v_xor per k-step and tile (a K loop is priced by its non-MFMA instructions: profiles/r04_mfma_issue.txt)
                constexpr int CGS = HALF == 4 ? 3 : 4;                      // (8 granules per input pixel <-> 4 k-steps per tap, 16 <-> 8)
                unsigned tapb[NT];
                auto pixels = [&](int k, h16x8 (&x)[NT]) {                  // k is a compile-time constant after unrolling
                    const int tap = k / HALF, j = k % HALF;
                    const int tap_off = (tap / 3) * p.IW + tap % 3;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if (j == 0) {
                            const int pix = lbase[nt] + tap_off;
                            tapb[nt] = lds_base + cur + (unsigned)(((pix << CGS) + (h ^ frame_swz(pix, CGS))) << 4);
                        }
                        x[nt] = __builtin_bit_cast(h16x8, *(lds_u4vp)(uintptr_t)(tapb[nt] ^ (unsigned)(32 * j)));
                    }
                };
                h16x8 xa[NT], xb[NT];
                pixels(0, xa);
#pragma unroll
                for (int k = 0; k < ksteps; ++k) {
                    const int d = k % R;
                    h16x8 (&xc)[NT] = (k & 1) ? xb : xa;
                    h16x8 (&xn)[NT] = (k & 1) ? xa : xb;
                    if (k + 1 < ksteps) pixels(k + 1, xn);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) acc[nt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, ring[d][nb]), xc[nt], acc[nt][nb], 0, 0, 0);
                    if (k + R < ksteps && TRS_FRAME_ABLATE != 1) {          // (compile-time) refill the slot with the k-step R ahead
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) ring[d][nb] = wload(nb);
                        wnext();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // epilogue: register quad q of block nb = channels cbase + nb*32 + 8q + 4h .. +3 of pixel r.  The two lanes of a pixel
                // (h = 0, 1) hold the two halves of every 8-channel group: they swap half of their quads (v_permlane32_swap_b32) so that
                // each lane ends with whole 16-byte groups — lane h = 0 stores groups q = 0, 1, lane h = 1 groups 2, 3
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    unsigned short* o = p.out + (size_t)(obase[nt] < 0 ? 0 : obase[nt]) * p.COUT + cbase;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        uint2 w[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float v0 = acc[nt][nb][4 * q], v1 = acc[nt][nb][4 * q + 1], v2 = acc[nt][nb][4 * q + 2], v3 = acc[nt][nb][4 * q + 3];
                            w[q] = p.relu ? relu_pack4(v0, v1, v2, v3) : make_uint2(pack_h16x2(v0, v1), pack_h16x2(v2, v3));
                        }
                        u4v g0, g1;
                        quad_groups(w, g0, g1);                           // lane h = 0: channel groups 0, 1 of this block, lane h = 1: groups 2, 3 (whole 16-byte groups)
#if TRS_FRAME_ABLATE == 3
                        asm volatile("" :: "v"(g0), "v"(g1)); (void)o;
#else
                        if (obase[nt] >= 0) {
                            *reinterpret_cast<u4v*>(o + nb * 32 + 16 * h) = g0;       // channels nb*32 + 16h .. + 7
                            *reinterpret_cast<u4v*>(o + nb * 32 + 16 * h + 8) = g1;   // ... + 8 .. + 15
                        }
#endif
                    }
                }
            }
        }
#if TRS_FRAME_STAMPS
        { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); fs_k += t_ - fs_t0; fs_t0 = t_; }
#endif
        __syncthreads();                                                    // the next group is in LDS, this one has been read
#if TRS_FRAME_STAMPS
        { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); fs_w += t_ - fs_t0; fs_t0 = t_; }
#endif
    }
#if TRS_FRAME_STAMPS
    if (blockIdx.x == 7 && lane == 0 && (wave == 0 || wave == COMPUTE - 1 || wave == COMPUTE))
        printf("frame kernel (cg %d, COUT %d, bands %d, F %d, %d groups), workgroup 7, wave %d [ticks]: first staging %lld | groups: work %lld, wait at the barrier %lld\n",
               p.cg, p.COUT, p.bands, p.F, n_groups, wave, fs_st, fs_k, fs_w);
#endif
}

// ---- conv4 .. conv7 in ONE launch: a frame's activations never leave LDS (round 2) ---------------------------------------------
// The time line of a workgroup of this loop (profiles/r02_pilot_dense.txt) shows ~3 us from dispatch to the first data in LDS and
// ~2.3 us from the last store to the end of the launch; the four 3x3 layers paid that four times over for 4-10 us of work each,
// plus a global round trip of every activation.  At 120x160 the four layers of F = 4 frames fit LDS together:
//   region A (76.8 KB): conv4's output -> (conv5 reads) -> conv6's output -> (conv7 reads)
//   region B (53.2 KB): conv4's input, two frames at a time -> conv5's output -> (conv6 reads)
// conv4 runs in two passes of two frames (its input is the largest tensor of the chain: 26 KB per frame); conv7 stores to global.
// A layer is trs_conv_frame_kernel's item loop unchanged (2 x 32 pixels x 2 x 32 channels per wave, weights straight from L2
// through a register ring, the K loop one basic block); its epilogue writes the next layer's LDS image — 16-byte granules in the
// same XOR-swizzled slots the staging DMA would have produced.  Layers are separated by a workgroup barrier only.
struct ChainLayer {
    const u4v* w; const float* bias;
    int IH, IW, OH, OW, COUT, cg, cgs;     // COUT == COUT_PAD (64 / 128); cg = input granules per pixel (8 / 16), cgs = log2
    int nt, nb;                            // a wave item = nt x 32 pixels x nb x 32 channels (nt 2 / 3, nb 1 / 2): the choice that leaves the busiest SIMD the fewest MFMAs
    unsigned magic_uout, magic_ow;         // floor(p / (OH x OW)) = umulhi(p, magic_uout), floor(p / OW) = umulhi(p, magic_ow) for p < 65536 (an item's set-up: four divisions
                                           // per tile cost ~100 vector instructions per item — a K loop is 36 k-steps of ~25)
};
struct ChainParams {
    const u4v* in;             // the first layer's input activation, fp16 NHWC
    unsigned short* out;       // the last layer's output activation
    int N, F, nl, split_first; // frames, frames per workgroup, layers (3 = conv5..7, 4 = conv4..7), first layer in two passes of F / 2 frames
    int offA, offB, off_bias;  // LDS byte offsets
    ChainLayer L[4];           // every layer but the last reads 64 channels (HALF = 4), the last 128 (HALF = 8)
};

#ifndef TRS_CHAIN_ABLATE
#define TRS_CHAIN_ABLATE 0   /* timing-only diagnostic builds of the chain's layers, never shipped: 1 = no weight refills, 2 = one LDS pixel read per item, 3 = no MFMA */
#endif
// One layer of the chain.  A wave item = NT x 32 pixels x NB x 32 output channels; the weight ring is R k-steps deep (R x NB granules per
// lane: NB = 1 takes twice the depth for the same registers and the same lead time in MFMA clocks).
template <int HALF, int R, int NT, int NB>
__device__ __forceinline__ void chain_layer(const ChainLayer& L, const u4v* lin, const float* lbias_f, int nu, u4v* lout, int cgs_out, int out_pix0,
                                            unsigned short* gout, int wave, int nwaves, int lane)
{
    const int r = lane & 31, h = lane >> 5;
    const int uout = L.OH * L.OW, m_wg = nu * uout;
    const int n_tiles = (m_wg + NT * 32 - 1) / (NT * 32), n_cgrp = L.COUT / (NB * 32);
    const float4* lbias = reinterpret_cast<const float4*>(lbias_f);
    constexpr int ksteps = 9 * HALF, CGS = HALF == 4 ? 3 : 4;              // (8 granules per input pixel <-> 4 k-steps per tap, 16 <-> 8: the host builds the layers so)
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<u4v*>(L.w), 0, 9 * L.cg * L.COUT * 16, 0x00020000);   // [9 taps x cg granules][COUT] granules
    const unsigned lin_b = (unsigned)(uintptr_t)lin;                                // LDS byte address of the layer's input image
    for (int item = wave; item < n_tiles * n_cgrp; item += nwaves) {
        const int cgrp = item / n_tiles, tile = item - cgrp * n_tiles;
        const int cbase = cgrp * NB * 32;
        int lbase[NT], mo[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int m = tile * NT * 32 + nt * 32 + r, mc = min(m, m_wg - 1);
            const int ul = (int)__umulhi((unsigned)mc, L.magic_uout), rem = mc - ul * uout, oy = (int)__umulhi((unsigned)rem, L.magic_ow), ox = rem - oy * L.OW;
            lbase[nt] = (ul * L.IH + oy) * L.IW + ox;
            mo[nt] = m < m_wg ? m : -1;
        }
        // (round 4, profiles/r04_mfma_issue.txt: inside a wave every vector-ALU instruction between two MFMAs ADDS ~6 ticks to the MFMA's 32, and two waves
        // per SIMD hide only part of each other's — a K loop is priced by its non-MFMA instructions per MFMA.  So: weights by buffer load (one lane offset
        // + a scalar per k-step instead of a 64-bit running pointer), and the XOR-swizzled LDS address of a tap's fragment computed ONCE per tap: the
        // granule of k-step j of the tap is 2 j + h, and (2 j + h) ^ swizzle = (h ^ swizzle) ^ 2 j — the tap's byte address ^ 32 j, one v_xor per k-step.)
        const int wvoff = (h * L.COUT + cbase + r) * 16;
        int wso = 0;                                                        // the k-step part of the offset: one scalar that runs with the loads (issued in k order)
        auto wload = [&](int nb) { return __builtin_bit_cast(u4v, __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff + nb * 512, wso, 0)); };
        auto wnext = [&]() { wso += 2 * L.COUT * 16; asm volatile("" : "+s"(wso)); };
        u4v ring[R][NB];
#pragma unroll
        for (int d = 0; d < R; ++d) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) ring[d][nb] = wload(nb);
            wnext();
        }
        f32x16 acc[NT][NB];
        acc_from_bias<NT, NB>(acc, lbias, cbase, h);
        [[maybe_unused]] h16x8 xkeep[NT];
        unsigned tapb[NT];                                                  // LDS byte address of this lane's granule h ^ swizzle of the current tap's pixel
        auto pixels = [&](int k, h16x8 (&x)[NT]) {
            const int tap = k / HALF, j = k % HALF;
            const int tap_off = (tap / 3) * L.IW + tap % 3;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (j == 0) {
                    const int pix = lbase[nt] + tap_off;
                    tapb[nt] = lin_b + (unsigned)(((pix << CGS) + (h ^ frame_swz(pix, CGS))) << 4);
                }
                const h16x8 v = __builtin_bit_cast(h16x8, *(lds_u4vp)(uintptr_t)(tapb[nt] ^ (unsigned)(32 * j)));
#if TRS_CHAIN_ABLATE == 2
                if (k == 0) xkeep[nt] = v;
                x[nt] = xkeep[nt];
#else
                x[nt] = v;
#endif
            }
        };
        h16x8 xa[NT], xb[NT];
        pixels(0, xa);
#pragma unroll
        for (int k = 0; k < ksteps; ++k) {
            const int d = k % R;
            h16x8 (&xc)[NT] = (k & 1) ? xb : xa;
            h16x8 (&xn)[NT] = (k & 1) ? xa : xb;
            if (k + 1 < ksteps) pixels(k + 1, xn);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
#if TRS_CHAIN_ABLATE == 3
                    asm volatile("" :: "v"(ring[d][nb]), "v"(xc[nt]));
#else
                    acc[nt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, ring[d][nb]), xc[nt], acc[nt][nb], 0, 0, 0);
#endif
                }
            if (k + R < ksteps && TRS_CHAIN_ABLATE != 1) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) ring[d][nb] = wload(nb);
                wnext();
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                uint2 w[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v0 = acc[nt][nb][4 * q], v1 = acc[nt][nb][4 * q + 1], v2 = acc[nt][nb][4 * q + 2], v3 = acc[nt][nb][4 * q + 3];
                    w[q] = relu_pack4(v0, v1, v2, v3);
                }
                u4v g0, g1;
                quad_groups(w, g0, g1);                                   // lane h = 0: channel groups 0, 1 of this block, lane h = 1: groups 2, 3 (whole 16-byte groups)
                if (mo[nt] >= 0) {
                    if (lout) {                                               // the next layer's image: pixel slot, granule ^ swizzle
                        const int po = out_pix0 + mo[nt], q0 = (cbase + nb * 32 + 16 * h) >> 3, sw = frame_swz(po, cgs_out);
                        lout[(po << cgs_out) + (q0 ^ sw)] = g0;
                        lout[(po << cgs_out) + ((q0 + 1) ^ sw)] = g1;
                    } else {
                        unsigned short* o = gout + (size_t)mo[nt] * L.COUT + cbase + nb * 32 + 16 * h;
                        *reinterpret_cast<u4v*>(o) = g0;
                        *reinterpret_cast<u4v*>(o + 8) = g1;
                    }
                }
            }
        }
    }
}
#ifndef TRS_CHAIN_R
#define TRS_CHAIN_R 4   /* weight ring depth of the chain's layers in k-steps at NB = 2 (twice that at NB = 1) */
#endif
// (tile height, channel blocks) of a layer -> its instantiation
template <int HALF>
__device__ __forceinline__ void chain_layer_any(const ChainLayer& L, const u4v* lin, const float* lbias_f, int nu, u4v* lout, int cgs_out, int out_pix0,
                                                unsigned short* gout, int wave, int nwaves, int lane)
{
    if (L.nt == 3) chain_layer<HALF, TRS_CHAIN_R, 3, 2>(L, lin, lbias_f, nu, lout, cgs_out, out_pix0, gout, wave, nwaves, lane);
    else chain_layer<HALF, TRS_CHAIN_R, 2, 2>(L, lin, lbias_f, nu, lout, cgs_out, out_pix0, gout, wave, nwaves, lane);
}
template <int BLOCK>
__global__ __launch_bounds__(BLOCK, 2) void trs_conv_chain_kernel(const ChainParams p)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int nwaves = BLOCK / 64;
    const int u0 = blockIdx.x * p.F, nu = min(p.F, p.N - u0);
    float* const lb = reinterpret_cast<float*>(psmem + p.off_bias);         // [layer][128]
    for (int li = 0; li < p.nl; ++li)
        for (int i = tid; i < p.L[li].COUT; i += BLOCK) lb[li * 128 + i] = p.L[li].bias[i];
    u4v* const A = reinterpret_cast<u4v*>(psmem + p.offA);
    u4v* const B = reinterpret_cast<u4v*>(psmem + p.offB);
    const unsigned lds0 = (unsigned)(uintptr_t)psmem;
    auto stage = [&](const ChainLayer& L, int fa, int cnt, unsigned lds_byte) {   // frames fa .. fa + cnt - 1 of L's input, whole and contiguous
        const int upix = L.IH * L.IW, total = (cnt * upix) << L.cgs;
        const u4v* src = p.in + ((size_t)fa * upix << L.cgs);
        for (int s0 = wave * 64; s0 < total; s0 += nwaves * 64) {
            const int sl = s0 + lane;
            if (sl < total) {
                const int pix = sl >> L.cgs, q = sl & (L.cg - 1);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (pix << L.cgs) + (q ^ frame_swz(pix, L.cgs))),
                                                 (__attribute__((address_space(3))) void*)(uintptr_t)(lds_byte + s0 * 16), 16, 0, 0);
            }
        }
    };
    int li = 0;
    const u4v* cur = A;
    if (p.split_first) {
        const ChainLayer& L0 = p.L[0];
        const int per = p.F / 2;
        for (int pass = 0; pass < 2; ++pass) {
            const int cnt = min(per, nu - per * pass);                      // workgroup-uniform
            if (cnt > 0) stage(L0, u0 + per * pass, cnt, lds0 + p.offB);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (cnt > 0) {
                chain_layer_any<4>(L0, B, lb, cnt, A, p.L[1].cgs, per * pass * L0.OH * L0.OW, nullptr, wave, nwaves, lane);
            }
            __syncthreads();
        }
        li = 1;
    } else {
        stage(p.L[0], u0, nu, lds0 + p.offA);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    for (; li < p.nl - 1; ++li) {
        const ChainLayer& L = p.L[li];
        u4v* const nxt = cur == A ? B : A;
        chain_layer_any<4>(L, cur, lb + li * 128, nu, nxt, p.L[li + 1].cgs, 0, nullptr, wave, nwaves, lane);
        __syncthreads();
        cur = nxt;
    }
    {
        const ChainLayer& L = p.L[p.nl - 1];
        chain_layer_any<8>(L, cur, lb + (p.nl - 1) * 128, nu, nullptr, 0, 0, p.out + (size_t)u0 * L.OH * L.OW * L.COUT, wave, nwaves, lane);
    }
}

// (Round 5 rebuilt this chain on v_mfma_f32_16x16x32_f16 with a conflict-free LDS image — two planes of channel granules, host tables that fill 16-pixel blocks by
// residue class: bank conflicts 46 % -> 9.5 % of the LDS cycles, MFMA cycles -15 % — and it was 7-14 % SLOWER on every box: the K loops are bound by the instructions
// around the MFMAs, and a 16x16x32 MFMA takes twice the issue slots per FLOP.  Removed again; profiles/r05_pilot_chain16.txt has the numbers and the stamps.)
// ---- conv3 with its input frames in LDS (round 2) -----------------------------------------------------------------------------
// conv3 (5x5 stride 2, 32 -> 64 channels) on the span kernel below reads three 1 KB fragments from LDS per two MFMAs (its weights
// live in LDS: 192 B/clk per CU at full MFMA rate against the 128 the LDS delivers) — 43 us per 1024 frames.  This is the frame
// kernel's scheme instead: F = 2 input frames (64 KB each) in LDS by LDS-DMA, weights straight from L2 through a register ring,
// 2 x 32 pixels x 2 x 32 channels per wave, the 50 k-steps (25 taps x 2 halves of the 32 input channels) one basic block.
// Stride 2 changes the LDS image: a row is stored as its even columns, then its odd columns, so the windows of consecutive
// output pixels are consecutive 64-byte slots for every tap; granule g of slot s sits at g ^ ((s >> 2) & 3) — 16 consecutive
// slots of one logical granule cover all 64 banks.
#ifndef TRS_F5_R
#define TRS_F5_R 4   /* weight ring depth of trs_conv_frame5_kernel in k-steps */
#endif
#ifndef TRS_F5_NB
#define TRS_F5_NB 2        /* 32-channel blocks per wave item */
#define TRS_F5_COMPUTE 4   /* compute waves (+ 4 loader waves); 1 x 8 and 2 x 8 measured: see the kernel */
#endif
#ifndef TRS_F5_STAMPS
#define TRS_F5_STAMPS 0   /* diagnostic build, never shipped: workgroup 7 prints the s_memtime ticks of its first staging, its work and its waits */
#endif
struct Frame5Params {
    const u4v* in;             // fp16 NHWC [N][IH][IW][32]
    const u4v* w;              // [KH*KW*4][64] granules: row (kh, kw, c8), 2 k + h = granule of k-step k, half h
    const float* bias;
    unsigned short* out;       // fp16 NHWC [N][OH][OW][64]
    int N, IH, IW, OH, OW, F, ev, COUT;   // ev = even columns per row = (IW + 1) / 2
    int bands, ohb, ihb;                  // a frame that does not fit LDS is cut into `bands` bands of ohb output rows = ihb = 2 ohb + 3 input rows (bands == 1: ohb = OH, ihb = IH)
    unsigned magic_ow;                    // floor(p / OW) = umulhi(p, magic_ow)
};

// Round 4: the workgroup is persistent and double-buffered.  Until then a 4-wave workgroup took ONE unit (two workgroups per CU, "one stages
// while the other computes") — but all workgroups of a launch start together, so both of a CU staged at the same time and then computed at the
// same time: stamps (-DTRS_F5_STAMPS=1) showed 10-12 k clocks of staging — 512 workgroups pulling 64 KB each at once: the fabric's ~6.5 TB/s —
// in front of 10-11 k clocks of K loops, twice over for the 1024 frames.  Now one 8-wave workgroup per CU walks its units: waves 0..3 compute unit i
// (one NT x 2 item each at 120x160) from one LDS buffer while waves 4..7 run the LDS-DMA of unit i + 1 into the other; one barrier per unit.
// conv3 33.5 -> 31.1 us per 1024 frames.  What is left (stamps of the persistent form, profiles/r04_pilot_frame5.txt): a unit's K loops take 10.5-11.6 k
// s_memtime ticks for 200 MFMAs per wave (5.6 k at the MFMA rate: ONE compute wave per SIMD keeps its pipe ~55 % busy; two waves per SIMD of the old form
// kept it full, but staged and computed in lockstep).  Timing-only builds: no weight refills -8 %, refills always from L1 -7 %, no pixel reads -2 % —
// neither operand stream is what holds a lone wave back.  Measured and not kept: 8 compute waves on 32-channel items (NB = 1: twice the LDS
// traffic) 32.4 us; 8 compute waves on the four NB = 2 items (half of them idle) 31.4; weight rings 8 and 12 k-steps deep (possible since the
// weights come by buffer load: 138 registers instead of 234) 32.3 / 32.5.
template <int NT, int NB, int R, int BLOCK, int COMPUTE>
__global__ __launch_bounds__(BLOCK, 1) void trs_conv_frame5_kernel(const Frame5Params p)
{
    constexpr int KW = 5, ksteps = 50, kCompute = COMPUTE;                  // waves 0 .. COMPUTE - 1 compute, the others stage
    const int COUT = p.COUT;                                                // 64 (the host checks)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int nwaves = BLOCK / 64;
    static_assert(nwaves > kCompute, "loader waves");
    const int r = lane & 31, h = lane >> 5;
    const int n_units = p.N * p.bands;
    const int upix = p.ihb * p.IW;                                          // input pixels staged per unit
    const unsigned buf_bytes = (unsigned)upix * 64u;                        // one unit in LDS (a multiple of 64)
    const unsigned lds_base = (unsigned)(uintptr_t)psmem;
#if TRS_F5_STAMPS
    long long f5_t0 = (long long)__builtin_amdgcn_s_memtime(), f5_st = 0, f5_k = 0, f5_w = 0;
#endif
    // staging of unit u into the buffer at byte offset `off` by waves w0 .. w0 + nw - 1: slot sl = 4 * lpix + q holds granule q ^ ((lpix >> 2) & 3) of the
    // pixel that lives in slot lpix
    auto stage = [&](int u, unsigned off, int w0, int nw) {
        const int total = upix * 4;
        const int f = u / p.bands, band = u - f * p.bands;
        for (int s0 = (wave - w0) * 64; s0 < total; s0 += nw * 64) {
            const int sl = s0 + lane;
            if (sl < total) {
                const int lpix = sl >> 2, g = (sl & 3) ^ ((lpix >> 2) & 3);
                const int iyl = lpix / p.IW, sx = lpix - iyl * p.IW;
                const int col = sx < p.ev ? 2 * sx : 2 * (sx - p.ev) + 1;
                const int iy = min(2 * band * p.ohb + iyl, p.IH - 1);         // (rows below the frame are never read by a valid pixel)
                const size_t gpix = ((size_t)f * p.IH + iy) * p.IW + col;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.in + gpix * 4 + g),
                                                 (__attribute__((address_space(3))) void*)(uintptr_t)(lds_base + off + s0 * 16), 16, 0, 0);
            }
        }
    };
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<u4v*>(p.w), 0, 100 * COUT * 16, 0x00020000);   // [25 taps x 4 granules][COUT] granules
    const int wvoff = (h * COUT + r) * 16;
    float* lb = reinterpret_cast<float*>(psmem + 2 * (size_t)buf_bytes);
    const float4* lbias = reinterpret_cast<const float4*>(lb);
    int u = blockIdx.x;
    if (u < n_units) stage(u, 0u, 0, nwaves);                               // the first unit: all waves
    for (int i = tid; i < COUT; i += BLOCK) lb[i] = p.bias[i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#if TRS_F5_STAMPS
    { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); f5_st += t_ - f5_t0; f5_t0 = t_; }
#endif
    for (int it = 0; u < n_units; ++it, u += gridDim.x) {
        const unsigned cur = (it & 1) ? buf_bytes : 0u;
        if (wave >= kCompute) {                                             // loaders: the next unit into the other buffer
            if (u + (int)gridDim.x < n_units) stage(u + (int)gridDim.x, buf_bytes - cur, kCompute, nwaves - kCompute);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            const int f = u / p.bands, band = u - f * p.bands;
            const int m_wg = min(p.ohb, p.OH - band * p.ohb) * p.OW;         // a short last band has no tiles for the rows below the frame
            const int n_tiles = (m_wg + NT * 32 - 1) / (NT * 32), n_cgrp = COUT / (NB * 32);
            for (int it2 = wave; it2 < n_tiles * n_cgrp; it2 += kCompute) {
                const int cgrp = it2 / n_tiles, item = it2 - cgrp * n_tiles, cbase = cgrp * NB * 32;   // (the waves of one SIMD take different tiles where they can)
                int lbase[NT];
                long long mo[NT];                                           // output pixel index in the layer's activation, -1 = no such pixel
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int m = item * NT * 32 + nt * 32 + r, mc = min(m, m_wg - 1);
                    const int oyl = (int)__umulhi((unsigned)mc, p.magic_ow), ox = mc - oyl * p.OW;
                    const int oy = band * p.ohb + oyl;
                    lbase[nt] = 2 * oyl * p.IW + ox;                        // slot of input pixel (2 oyl, 2 ox) of the unit: even plane, position ox
                    mo[nt] = m < m_wg ? ((long long)f * p.OH + oy) * p.OW + ox : -1;
                }
                // weight granule (2 k + h, cout nb*32 + r) by buffer load: ONE lane offset + a scalar per (k-step, block) — a 64-bit address per load
                // (or a running pointer per ring slot) cost this kernel ~80 registers, and with them a ring deep enough to cover an L2 hit
                u4v ring[R][NB];
                int wso = cbase * 16;                                       // the k-step part of the offset: one scalar that runs with the loads (see trs_conv_frame_kernel)
                auto wload = [&](int nb) { return __builtin_bit_cast(u4v, __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff + nb * 512, wso, 0)); };
                auto wnext = [&]() { wso += 2 * COUT * 16; asm volatile("" : "+s"(wso)); };
#pragma unroll
                for (int d = 0; d < R; ++d) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) ring[d][nb] = wload(nb);
                    wnext();
                }
                f32x16 acc[NT][NB];
                acc_from_bias<NT, NB>(acc, lbias, cbase, h);
                // (the swizzled address once per tap, ^ 32 for its second k-step: see trs_conv_frame_kernel)
                unsigned tapb[NT];
                auto pixels = [&](int k, h16x8 (&x)[NT]) {                  // k is a compile-time constant after unrolling
                    const int tap = k / 2, j = k % 2;
                    const int kh = tap / KW, kw = tap % KW;
                    const int tap_off = kh * p.IW + ((kw & 1) ? p.ev : 0) + (kw >> 1);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if (j == 0) {
                            const int pix = lbase[nt] + tap_off;
                            tapb[nt] = lds_base + cur + (unsigned)((pix * 4 + (h ^ ((pix >> 2) & 3))) << 4);
                        }
                        x[nt] = __builtin_bit_cast(h16x8, *(lds_u4vp)(uintptr_t)(tapb[nt] ^ (unsigned)(32 * j)));
                    }
                };
                h16x8 xa[NT], xb[NT];
                pixels(0, xa);
#pragma unroll
                for (int k = 0; k < ksteps; ++k) {
                    const int d = k % R;
                    h16x8 (&xc)[NT] = (k & 1) ? xb : xa;
                    h16x8 (&xn)[NT] = (k & 1) ? xa : xb;
                    if (k + 1 < ksteps) pixels(k + 1, xn);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) acc[nt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, ring[d][nb]), xc[nt], acc[nt][nb], 0, 0, 0);
                    if (k + R < ksteps) {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) ring[d][nb] = wload(nb);
                        wnext();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    unsigned short* o = p.out + (size_t)(mo[nt] < 0 ? 0 : mo[nt]) * COUT + cbase;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        uint2 w[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float v0 = acc[nt][nb][4 * q], v1 = acc[nt][nb][4 * q + 1], v2 = acc[nt][nb][4 * q + 2], v3 = acc[nt][nb][4 * q + 3];
                            w[q] = relu_pack4(v0, v1, v2, v3);
                        }
                        u4v g0, g1;
                        quad_groups(w, g0, g1);                           // lane h = 0: channel groups 0, 1 of this block, lane h = 1: groups 2, 3 (whole 16-byte groups)
                        if (mo[nt] >= 0) {
                            *reinterpret_cast<u4v*>(o + nb * 32 + 16 * h) = g0;
                            *reinterpret_cast<u4v*>(o + nb * 32 + 16 * h + 8) = g1;
                        }
                    }
                }
            }
        }
#if TRS_F5_STAMPS
        { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); f5_k += t_ - f5_t0; f5_t0 = t_; }
#endif
        __syncthreads();                                                    // the next unit is in LDS, this one has been read
#if TRS_F5_STAMPS
        { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); f5_w += t_ - f5_t0; f5_t0 = t_; }
#endif
    }
#if TRS_F5_STAMPS
    if (blockIdx.x == 7 && lane == 0 && (wave == 0 || wave == kCompute - 1 || wave == kCompute))
        printf("frame5, workgroup 7, wave %d [clocks]: first staging %lld | units: work %lld, wait at the barrier %lld\n", wave, f5_st, f5_k, f5_w);
#endif
}

// conv1 -> conv2 fused: conv1's activation (217 KB per 120x160 frame, the largest tensor of the network: 222 MB per 1024
// frames, written once and read once) never leaves the CU.  A workgroup takes a band of R2 conv2 output rows of one
// frame, computes the 2 R2 + 3 conv1 rows it needs into an LDS tile (same bias + ReLU + fp16 rounding as the unfused
// layer, so conv2's result is bit-identical), and runs conv2 on that tile with its fragments read by ds_read_b128.
// Neighbouring bands recompute 3 conv1 rows each ((2 R2 + 3) / (2 R2) of the conv1 work).
// Measured and dropped (1024 frames of 120x160; this form: 128-131 us):
//   * two wave teams and two tiles (conv1 of the next item beside conv2 of the current one): 8 / 10 / 12 of the 16 waves on
//     conv1 -> 209 / 180 / 160 us - conv1 scales with the waves it gets;
//   * two conv1 tiles requested together per wave (36 dwords in flight): 141 us;
//   * ONE unaligned 8-byte load per k-step instead of three aligned dwords + v_alignbyte: 144 us (267 -> 297 us at 240x320) -
//     the misaligned 8-byte loads cost the addresser more than the three aligned ones.
#ifndef TRS_FUSE_ABLATE
#define TRS_FUSE_ABLATE 0
#endif
#ifndef TRS_C2_ABLATE
#define TRS_C2_ABLATE 0
#endif
#ifndef TRS_C2_NT
#define TRS_C2_NT 1      /* conv2 tiles of the fused head per wave item (one weight fragment feeds that many MFMAs); 2 measured: see the kernel */
#endif
#ifndef TRS_C2_DEPTH
#define TRS_C2_DEPTH 4   /* k-steps of fragments in flight */
#endif
#ifndef TRS_BAND_STAMPS
#define TRS_BAND_STAMPS 0   /* diagnostic build, never shipped: workgroup 7's waves 0 and 8 add up the shader clocks of their phases and print them behind a last barrier */
#endif
#if TRS_BAND_STAMPS
#define BAND_STAMP(i) do { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); st_[i] += t_ - t0_; t0_ = t_; } while (0)
#else
#define BAND_STAMP(i) do { } while (0)
#endif
struct Fuse12Params {
    const uint8_t* frames; int frames_bytes;
    const u4v* w1; const float* b1;                        // conv1: [12][32] granules, [32]
    const u4v* w2;                                          // conv2: [80][32] granules (kernel rows padded 15 -> 16)
    ConvParams c2;                                          // conv2's output side (out, COUT, relu, nt_out, bias)
    int N, IH, IW, OH1, OW1, OH2, OW2, R2, bands;
    int off_w2, off_b, off_tile;                            // LDS layout
    int tile_bytes;
    int off_band, band_bytes;                               // band kernel: the frame rows under the conv1 tile as fp16 [rows][IW * 3]
    int wsplit, w2p, cpr;                                   // band kernel cut in width: parts per band, conv2 columns per part, 16-byte chunks per staged row
    int roll;                                               // band kernel: a workgroup walks the bands of a (frame, part) top to bottom and keeps the 3 shared conv1 rows (see the kernel)
    int c1_bounded;                                         // conv1's sums provably stay inside binary16 (|bias| + sum |w| < 65504, pixels / 256 <= 1): its epilogue skips the saturation
    unsigned magic_full, magic_cpr;                         // floor(p / w1) = umulhi(p, magic_full) for a part's conv1 width 2 w2p + 3; the same for / cpr
};

// The fused head with conv1's input staged once per band.  (Rounds 1-2 had a direct form whose conv1 tiles fetched their windows
// straight from the frame: 18 scattered dword loads and ~100 VALU ops per lane and 32-pixel tile for 6 MFMAs, each frame byte fetched and
// unpacked ~6 times — conv1's phase was 70 of 131 us, bound by the texture addresser; removed in round 4, a frame shape whose band does not
// fit runs the two layers unfused.)  Here the workgroup loads the band's frame rows ONCE (contiguous in the frame: one coalesced 16-byte load per
// thread, requested one item ahead so that its latency hides behind conv1 of the current item), unpacks them once into a fp16
// image in LDS, and conv1 reads its k-steps (8 consecutive values, 4-byte aligned) from there with two ds_read2_b32.
// Same fp16 values as the separate layers; conv2's k dimension runs in column-parity order (see the tile layout below).
constexpr int kBandPf = 2;                                                  // 16-byte chunks of the band per loader thread (waves 8..15: 512 threads; 4 until round 2:
                                                                            // the 8 registers now hold conv1's bias)
template <bool SPLIT>
__global__ __launch_bounds__(1024) void trs_conv12_band_kernel(const Fuse12Params q)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    u4v* lw1 = reinterpret_cast<u4v*>(psmem);                              // [12][32]
    u4v* lw2 = reinterpret_cast<u4v*>(psmem + q.off_w2);                   // [80][32]
    float4* lb1 = reinterpret_cast<float4*>(psmem + q.off_b);              // [8]
    float4* lb2 = lb1 + 8;                                                 // [8]
    // conv1 tile in LDS: [r1 rows][2 planes: even / odd conv1 columns][plane_px][24] fp16.  conv2 (stride 2) reads, per output
    // pixel x2 and kernel row, the conv1 columns 2 x2 .. 2 x2 + 4: with the columns interleaved the 16 lanes of a ds_read_b128
    // group sit 96 bytes apart — an EVEN number of 16-byte slots, so only 8 of the 16 bank groups are hit (2-way conflict on
    // every fragment read, and no padding changes the parity).  Split by column parity, consecutive x2 are 48 bytes = 3 slots
    // apart (odd: all 16 bank groups), and a window is two runs: even columns (3 pixels = 9 slots) then odd (2 pixels = 6
    // slots) — conv2's granules are packed in that order for this kernel (w2 = the handle's parity-ordered copy).
    unsigned char* tile1 = psmem + q.off_tile;
    const int plane_px = SPLIT ? q.w2p + 2 : (q.OW1 + 1) >> 1, plane_bytes = plane_px * 48, tile_pitch = 2 * plane_bytes;   // (a part's conv1 width is 2 w2p + 3)
    unsigned char* band = psmem + q.off_band;                              // [2 r1 + 3][IW * 3] fp16 (+ padding)
    for (int i = tid; i < 12 * 32; i += blockDim.x) lw1[i] = q.w1[i];
    for (int i = tid; i < 80 * 32; i += blockDim.x) lw2[i] = q.w2[i];
    for (int i = tid; i < 8; i += blockDim.x) { lb1[i] = *reinterpret_cast<const float4*>(q.b1 + 4 * i); lb2[i] = *reinterpret_cast<const float4*>(q.c2.bias + 4 * i); }
    for (int i = tid; i < q.tile_bytes / 16; i += blockDim.x) reinterpret_cast<u4v*>(tile1)[i] = (u4v)(0u);   // never multiply an uninitialised bit pattern by a zero weight
    for (int i = tid; i < q.band_bytes / 16; i += blockDim.x) reinterpret_cast<u4v*>(band)[i] = (u4v)(0u);

    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(q.frames), 0, q.frames_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout2 = __builtin_amdgcn_make_buffer_rsrc(static_cast<unsigned char*>(q.c2.out), 0, q.c2.M * 64, 0x00020000);   // conv2's activation: 32 couts
    const float inv_ow2 = 1.0f / (float)q.OW2;
    constexpr bool kC2NtIsOne = TRS_C2_NT == 1;
    const int row_in = q.IW * 3;                                            // bytes per frame row
    const int bpitch = SPLIT ? q.cpr * 16 : row_in;                         // values per row of the band image
    auto divmod = [](int v, int d, float inv, int& qt, int& rm) {
        qt = (int)(((float)v + 0.5f) * inv); rm = v - qt * d;
        if (rm < 0) { --qt; rm += d; } else if (rm >= d) { ++qt; rm -= d; }
    };
    // The workgroup's item `it` = (frame n, band of R2 conv2 rows from y2_0[, part of w2 conv2 columns from x2_0]); w1 = the part's conv1 columns.
    // Two orders (q.roll, chosen by the host):
    //   flat     item blockIdx.x + it * gridDim.x of all (frame, band, part) triples - neighbouring bands go to different workgroups, every band
    //            computes all 2 r2 + 3 conv1 rows under it (3 of them a second time);
    //   rolling  (round 3) the workgroup takes whole (frame, part) streams and walks their bands top to bottom: the conv1 tile is a ring of
    //            NR = 2 R2 + 3 rows (conv1 row y lives in slot y mod NR), the last 3 rows of the previous band are still in it, so a band after
    //            the first computes only its 2 r2 NEW rows (`skip` = 3) from a band image that starts 6 frame rows lower: a fifth less conv1
    //            arithmetic, unpacking and LDS traffic, and 30 instead of 37 conv1 tiles per band at 120x160 (two rounds of the 16 waves, not
    //            three).  Same values into the same MFMAs: bit-identical to the flat order.
    const int NR = 2 * q.R2 + 3;
    auto geometry = [&](int it, int& n, int& y2_0, int& r2, int& r1, int& x2_0, int& w2, int& w1, int& skip) {
        int b, part = 0;
        if (q.roll) {
            const int sidx = it / q.bands;
            b = it - sidx * q.bands;
            const int stream = (int)blockIdx.x + sidx * (int)gridDim.x;
            if constexpr (SPLIT) { n = stream / q.wsplit; part = stream - n * q.wsplit; } else n = stream;
            skip = b > 0 ? 3 : 0;
        } else {
            const int wt = (int)blockIdx.x + it * (int)gridDim.x;
            if constexpr (SPLIT) {
                const int per = q.bands * q.wsplit;
                n = wt / per;
                const int rem = wt - n * per;
                b = rem / q.wsplit; part = rem - b * q.wsplit;
            } else {
                n = wt / q.bands;
                b = wt - n * q.bands;
            }
            skip = 0;
        }
        y2_0 = b * q.R2;
        if constexpr (SPLIT) { x2_0 = min(part * q.w2p, q.OW2 - q.w2p); w2 = q.w2p; w1 = 2 * w2 + 3; }   // every part w2p columns wide: the last one starts where it ends at the frame's edge and
                                                                                                       // recomputes the column(s) it shares with its neighbour (the same values to the same bytes)
        else { x2_0 = 0; w2 = q.OW2; w1 = q.OW1; }
        r2 = min(q.R2, q.OH2 - y2_0); r1 = 2 * (r2 - 1) + 5;
    };
    // the band of item wt: frame rows 4 y2_0 .. + 2 r1 + 2 — whole rows are contiguous in the frame; a part takes 16 cpr bytes of each
    // row from column 4 x2_0 on (what it reads past its own 2 w1 + 3 pixels is never used)
    // A loader thread moves HALF chunks (8 frame bytes -> 16 bytes of the image): consecutive lanes then write consecutive 16-byte
    // slots (whole chunks per lane put the two ds_write_b128 of a lane 32 bytes apart: a 2-way bank conflict on every write)
    auto request = [&](int it, uint2 (&raw)[2 * kBandPf]) {
        int n, y2_0, r2, r1, x2_0, w2, w1, skip;
        geometry(it, n, y2_0, r2, r1, x2_0, w2, w1, skip);
        const int row0 = 4 * y2_0 + 2 * skip, rows = 2 * (r1 - skip) + 3;   // the frame rows under the conv1 rows this item computes
        int tl = tid - 512;                                                  // this loader thread; opaque to the optimiser: the per-chunk indices below are
        asm volatile("" : "+v"(tl));                                         // a few shifts per band, not eight registers held (and spilled) through both phases
        if constexpr (SPLIT) {
            const int start = ((n * q.IH + row0) * q.IW + 4 * x2_0) * 3, nchunk = rows * q.cpr;
#pragma unroll
            for (int j = 0; j < 2 * kBandPf; ++j) {
                const int hc = tl + j * 512, c = hc >> 1, row = (int)__umulhi((unsigned)max(c, 0), q.magic_cpr), kk = c - row * q.cpr;
                raw[j] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rin, c < nchunk ? start + row * row_in + 16 * kk + 8 * (hc & 1) : q.frames_bytes, 0, 0));   // past the end: zeros
            }
        } else {
            const int start = (n * q.IH + row0) * row_in, nchunk = (rows * row_in) >> 4;
#pragma unroll
            for (int j = 0; j < 2 * kBandPf; ++j) {
                const int hc = tl + j * 512;
                raw[j] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rin, (hc >> 1) < nchunk ? start + 8 * hc : q.frames_bytes, 0, 0));   // past the end: zeros
            }
        }
    };
    auto unpack = [&](const uint2 (&raw)[2 * kBandPf]) {                     // 8 uint8 -> 8 fp16 (exact), 16 bytes of the band image
#pragma unroll
        for (int j = 0; j < 2 * kBandPf; ++j) {
            const int hc = (tid - 512) + j * 512;
            if (hc * 16 + 16 > q.band_bytes) continue;
            // two pixels as x / 256 in binary16 (exact: 0, 2^-8 .. 255 x 2^-8) in TWO instructions: binary16's 4.0 (0x4400) has an ulp of 2^-8, so
            // 0x4400 | x IS 4 + x / 256; a byte permute drops the two bytes into the mantissas of (4.0, 4.0) and one packed subtract takes the 4.0 away
            // (until round 4: v_cvt_f32_ubyte x 2, v_cvt_pkrtz, v_pk_mul_f16 — the loader waves share their SIMDs with the conv2 waves, and every
            // vector instruction there is time the MFMAs do not get, profiles/r04_mfma_issue.txt)
            auto pair = [](unsigned w, int k) -> unsigned {
                typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
                const unsigned sel = k == 0 ? 0x05010500u : 0x05030502u;     // result bytes, low to high: w.byte[k], 0x44, w.byte[k + 1], 0x44 (sources 0..3 = w, 4..7 = the magic)
                const unsigned m = __builtin_amdgcn_perm(0x44004400u, w, sel);   // = the halves 0x4400 | w.byte[k], 0x4400 | w.byte[k + 1]
                const h16x2 four = {(_Float16)4.0f, (_Float16)4.0f};
                return __builtin_bit_cast(unsigned, __builtin_bit_cast(h16x2, m) - four);
            };
            *reinterpret_cast<u4v*>(band + (size_t)hc * 16) = u4v{pair(raw[j].x, 0), pair(raw[j].x, 2), pair(raw[j].y, 0), pair(raw[j].y, 2)};
        }
    };

    // Waves 8..15 are the loaders: they have no conv2 tiles (and so no stores in their memory queue to wait behind), request a
    // band a whole item ahead and unpack it while waves 0..7 run conv2.
    const int n_streams = q.N * (SPLIT ? q.wsplit : 1), n_flat = n_streams * q.bands;
    const int mine = q.roll ? n_streams : n_flat;                           // what the workgroups share out: streams of bands, or single items
    const int total = ((int)blockIdx.x < mine ? (mine - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0) * (q.roll ? q.bands : 1);   // this workgroup's items
    const bool loader = wave >= 8;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);                // tile indices in scalar registers
    const unsigned magic1 = (unsigned)((0x100000000ull + (unsigned)q.OW1 - 1u) / (unsigned)q.OW1);   // floor(p / OW1) = umulhi(p, magic1) for p < 65536
    int wt = 0;                                                             // this workgroup's item counter
    uint2 raw[2 * kBandPf];
    if (loader && wt < total) request(wt, raw);
    __syncthreads();                                                        // weights staged, tile and band zeroed
    if (loader && wt < total) {
        unpack(raw);
        if (wt + 1 < total) request(wt + 1, raw);                           // the second item's band is on its way
    }
    __syncthreads();
    // conv1's weights are the same for every tile of every band: this lane's five granules stay in registers for the whole kernel (they were 6 of the
    // 11 LDS reads of a conv1 tile; re-reading them per band — two dependent LDS round trips in front of every band's first tile — was measured in
    // round 4: ~0.5 k of a band's 4 k clocks of phase 1).  So does the bias: it is the first MFMA's C operand (16 registers: couts 8 qd + 4 h + j in
    // register 4 qd + j, zeros for the padding couts 24..31), and the band image holds the pixels as x / 256 (exact), so the sums need no 2^-8 and
    // no bias FMA.
    u4v wv[5];
#pragma unroll
    for (int s6 = 0; s6 < 5; ++s6) wv[s6] = lw1[(2 * s6 + h) * 32 + r];
#pragma unroll
    for (int s6 = 0; s6 < 5; ++s6) asm volatile("" : "+v"(wv[s6]));         // keep them in registers: do not re-read them per tile
    f32x16 bias16;
#pragma unroll
    for (int qd = 0; qd < 3; ++qd) {
        const float4 bb = lb1[2 * qd + h];
        bias16[4 * qd] = bb.x; bias16[4 * qd + 1] = bb.y; bias16[4 * qd + 2] = bb.z; bias16[4 * qd + 3] = bb.w;
    }
#pragma unroll
    for (int i = 12; i < 16; ++i) bias16[i] = 0.0f;
    asm volatile("" : "+v"(bias16));
    // the rolling bands of whole-width frames share one conv1 geometry (2 R2 new rows of OW1 pixels: at most two tiles per wave): this lane's
    // window address in the band image, its column's offset in a ring row, and (pixel index | row << 16) for the tiles wave and wave + 16
    // ... and so do their conv2 tiles (one per wave, kC2Nt = 1): the tile pixel's column address in a ring row, and (row << 24 | byte offset of its 32 output
    // bytes relative to the band's first output pixel); a lane past a full band's last pixel takes that pixel (never stored)
    const int w2c = SPLIT ? q.w2p : q.OW2, w1c = SPLIT ? 2 * q.w2p + 3 : q.OW1;      // conv2 / conv1 columns of an item (the same for every item)
    const bool c2_fast = kC2NtIsOne && (q.R2 * w2c + 31) / 32 <= 8 && q.R2 * q.OW2 * 64 < (1 << 24);
    unsigned c2s_abase = 0, c2s_rel = 0;
    if (c2_fast) {
        const int mm = min(wave_u * 32 + r, q.R2 * w2c - 1);
        const int yl2 = (int)(((float)mm + 0.5f) * (SPLIT ? 1.0f / (float)w2c : inv_ow2));
        int yq = yl2, x2 = mm - yl2 * w2c;
        if (x2 < 0) { --yq; x2 += w2c; } else if (x2 >= w2c) { ++yq; x2 -= w2c; }
        c2s_abase = (unsigned)q.off_tile + (unsigned)__mul24(x2, 48);
        c2s_rel = ((unsigned)yq << 24) | (unsigned)((__mul24(yq, q.OW2) + x2) * 64 + 32 * h);
    }
    struct C1Slot { unsigned ra, pk; };                                     // pk = pixel index (10 bits) | row (4 bits) | column offset in a ring row (the rest)
    C1Slot c1s[2];
    const bool c1_fast = q.roll && (2 * q.R2 * w1c + 31) / 32 <= 2 * nwaves && 2 * q.R2 < 16 && 2 * nwaves * 32 <= 1024;
    if (c1_fast) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pp = (wave_u + i * nwaves) * 32 + r;
            const int yl = (int)__umulhi((unsigned)pp, SPLIT ? q.magic_full : magic1), x = pp - yl * w1c;
            c1s[i].ra = (unsigned)q.off_band + 16u * h + (unsigned)(__mul24(2 * yl, bpitch) + __mul24(x, 6)) * 2u;   // (past the band's last pixel: still inside the image
                                                                                                                  // buffer, which is sized for a frame's first band; never stored)
            c1s[i].pk = (unsigned)pp | ((unsigned)yl << 10) | ((unsigned)(((x & 1) ? plane_bytes : 0) + __mul24(x >> 1, 48)) << 14);
        }
    }
#if TRS_BAND_STAMPS
    long long st_[5] = {0, 0, 0, 0, 0}, t0_ = (long long)__builtin_amdgcn_s_memtime();
#endif
    while (wt < total) {
        const int nxt = wt + 1;                                             // uniform per workgroup
        int n, y2_0, r2, r1, x2_0, w2, w1, skip;
        geometry(wt, n, y2_0, r2, r1, x2_0, w2, w1, skip);
        const unsigned magic = SPLIT ? q.magic_full : magic1;
        const int s0 = q.roll ? (2 * y2_0) % NR : 0;                        // ring slot of the band's first conv1 row
        // ---- phase 1: the band's conv1 rows that are not in the tile yet (all of them, or all but the first 3), from the fp16 image ----
        const int rows1 = r1 - skip, npx1 = rows1 * w1, ntile1 = (npx1 + 31) >> 5;
#if TRS_FUSE_ABLATE != 1
        // one conv1 tile: five fragments from the band image (kernel rows 0..4 of this lane's window, half h), five MFMAs (the bias is the first one's C
        // operand), ReLU + fp16, three 8-byte writes into the ring (couts 8 qd + 4 h .. + 3, qd = 0..2: 24 channels)
        // (8-byte writes: the two column planes share bank groups, a 2-way conflict.  Swapping halves between the two lanes of a pixel (v_permlane32_swap_b32)
        // to write whole 16-byte granules was measured: head 100.7 -> 103.5 us on one box — the exchange and its selects cost more than the conflict.
        // Requesting the next tile's fragments right behind this tile's MFMAs was measured too: the oldest wave of a SIMD finishes 5 % earlier, the phase —
        // which ends with the youngest — not at all: head 76.0 us against 75.9.)
        auto conv1_tile = [&](unsigned rd, bool mine, unsigned dst_off) {
            u4v xv[5];                                                      // every fragment of the tile first: one LDS round trip, then the MFMAs back to back
#pragma unroll
            for (int s6 = 0; s6 < 5; ++s6) {
                const unsigned* src = reinterpret_cast<const unsigned*>(psmem + (rd + (unsigned)(s6 * bpitch * 2)));
                xv[s6] = u4v{src[0], src[1], src[2], src[3]};
            }
            __builtin_amdgcn_sched_barrier(0);                              // (left alone, hipcc interleaves the reads with the MFMAs two deep to save registers)
            f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, wv[0]), __builtin_bit_cast(h16x8, xv[0]), bias16, 0, 0, 0);
#pragma unroll
            for (int s6 = 1; s6 < 5; ++s6)                                  // (the sixth k-step of the padded weight layout is all zeros: skipped, + 0 changes nothing)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, wv[s6]), __builtin_bit_cast(h16x8, xv[s6]), acc, 0, 0, 0);
            if (mine) {
                uint2* dst = reinterpret_cast<uint2*>(tile1 + dst_off);
                if (q.c1_bounded) {                                         // (wave-uniform: conv1 cannot leave binary16's range with these weights — no saturation step)
#pragma unroll
                    for (int qd = 0; qd < 3; ++qd) dst[2 * qd + h] = relu_pack4_bounded(acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]);
                } else {
#pragma unroll
                    for (int qd = 0; qd < 3; ++qd) dst[2 * qd + h] = relu_pack4(acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]);
                }
            }
        };
        if (c1_fast && skip == 3) {
            // a band after the first of its frame (rolling): 2 R2 new rows, the same geometry for every such band — this lane's windows and columns were
            // worked out once (c1s), only the ring slot of its row moves with the band
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (wave_u + i * nwaves < ntile1) {                         // (wave-uniform)
                    const int ppi = (int)(c1s[i].pk & 1023u), yli = (int)((c1s[i].pk >> 10) & 15u);
                    int slot1 = s0 + 3 + yli;                                // (< 2 NR)
                    slot1 = slot1 >= NR ? slot1 - NR : slot1;
                    conv1_tile(c1s[i].ra, ppi < npx1, (unsigned)__mul24(slot1, tile_pitch) + (c1s[i].pk >> 14));
                }
            }
        } else {
            // the general walk (a frame's first band, bands cut in width, one band per item): a lane walks its pixels p = 32 t1 + r, t1 = wave, wave + 16, ...
            // incrementally: (row yl, column x), the LDS address of its window in the band image and its ring slot in the tile advance by wave-uniform steps
            // with one wrap
            const int step = nwaves * 32;
            const int adv_y = (int)__umulhi((unsigned)step, magic), adv_x = step - adv_y * w1;          // step = adv_y rows + adv_x columns
            const int adv_slot = adv_y % NR;
            int pp = wave_u * 32 + r;
            int yl = (int)__umulhi((unsigned)pp, magic), x = pp - yl * w1;
            const unsigned band_off = (unsigned)q.off_band + 16u * h;
            unsigned ra = band_off + (unsigned)(__mul24(2 * yl, bpitch) + __mul24(x, 6)) * 2u;             // kernel row 0 of this lane's window, half h
            const unsigned ra_last = band_off + (unsigned)(__mul24(2 * (rows1 - 1), bpitch) + (w1 - 1) * 6) * 2u;   // lanes past the band's last pixel compute it again (never stored)
            const unsigned ra_step = (unsigned)(adv_y * 2 * bpitch + adv_x * 6) * 2u, ra_wrap = (unsigned)(2 * bpitch - w1 * 6) * 2u;
            int slot1 = s0 + skip + yl;                                      // (< 2 NR)
            slot1 = slot1 >= NR ? slot1 - NR : slot1;
            for (int t1 = wave_u; t1 < ntile1; t1 += nwaves) {
                conv1_tile(min(ra, ra_last), pp < npx1, (unsigned)(__mul24(slot1, tile_pitch) + ((x & 1) ? plane_bytes : 0) + __mul24(x >> 1, 48)));
                pp += step; x += adv_x; slot1 += adv_slot; ra += ra_step;
                if (x >= w1) { x -= w1; slot1 += 1; ra += ra_wrap; }
                slot1 = slot1 >= NR ? slot1 - NR : slot1;
            }
        }
#endif
        BAND_STAMP(0);                                                      // conv1 tiles
        __syncthreads();                                                    // the tile is complete, the band image is free
        BAND_STAMP(1);                                                      // wait at the barrier
        if (loader) {
            if (nxt < total) {
                unpack(raw);                                                // the next item's band (requested an item ago)
                if (nxt + 1 < total) request(nxt + 1, raw);
            }
        } else {
            // ---- phase 2: conv2 rows from the tile (waves 0..7: at most a handful of tiles per band) ----
            const int npx2 = r2 * w2, ntile2 = (npx2 + 31) >> 5;
            const int m0 = (n * q.OH2 + y2_0) * q.OW2 + x2_0;               // first output pixel of the band (a whole band is consecutive in memory)
            const float inv_w2 = SPLIT ? 1.0f / (float)w2 : inv_ow2;
#if TRS_FUSE_ABLATE != 2
            // Phase stamps of round 4 (-DTRS_BAND_STAMPS=1, profiles/r04_pilot_head_stamps.txt; clocks per band of 7 tiles at 120x160):
            //   * The K loop is bound by LDS bandwidth: every MFMA takes 2 KB out of LDS (its weight and its pixel fragment), 560 KB per band =
            //     4.4 k clocks at 128 B/clk; the waves of a SIMD finish in age order, the phase ends ~4.4 k clocks after it began.
            //   * Until round 4 the epilogue went through a wave-private LDS stage (four broadcast bias reads, then the transpose: six dependent LDS
            //     round trips of ~200 clocks each while the other waves saturate the LDS): 1.4 k clocks per tile.  Now the accumulators start at the
            //     bias and the epilogue is ReLU + fp16 + v_permlane32_swap_b32 + two 16-byte stores per lane, no LDS: ~0.6 k.  Head 95.9 -> 88.5 us.
            //   * TRS_C2_NT = 2 (a wave takes a PAIR of tiles, one weight fragment feeds two MFMAs: 480 KB per band; fragments kC2Depth k-steps ahead
            //     by hand): the pairs run on four waves, one per SIMD, and each needs 3.7 k clocks for its 80 MFMAs + 1.0 k for two epilogues — 95.7 us
            //     on the same box.  (Round 2 measured the same split with compiler-scheduled reads: 107 -> 130 us; conv2's 40 weight fragments in
            //     registers on eight waves of 256 registers 100 -> 117 us, the same with K split over two waves 100 -> 125 us, r02_pilot_head_reg.txt.)
            //   * s_setprio 3 while the loader waves unpack (measured, removed: they finish in 1.0 k instead of 3.0 k clocks, but the conv2 waves
            //     they displace are the critical path): 90.0 us.  Deeper fragment rings (6, 8 k-steps): no change.
            //   * The odd k-steps' weight fragments straight from global memory (buffer loads, three in flight = six k-steps of lead; the LDS then delivers
            //     1.5 KB per MFMA and the L1 the rest): K loop 2.5 k -> 3.5 k clocks per band, head 76.3 -> 81 us (240x320: 162 -> 191) — the frames the
            //     loader waves stream through the same 32 KB L1 evict the weights, and an L2 hit is longer than the lead 128 registers leave room for.
            constexpr int kC2Nt = TRS_C2_NT, kC2Depth = TRS_C2_DEPTH;
            // window slot sl = 2 ks + h of a kernel row (ks = k-step within the row): 0..8 = the even run, 9..14 = the odd run, 15 = padding (zero
            // weights: the odd run's next 16 bytes).  Per lane that is three bases (+ 32 ks as the instruction's immediate offset):
            const unsigned off_lo = 16 * h;                                              // ks 0..3: slots 0..7
            const unsigned off_mid = h ? (unsigned)plane_bytes : 128u;                   // ks 4: slot 8 (even run) / 9 (odd run, first)
            const unsigned off_hi = (unsigned)plane_bytes - 144u + 16 * h;               // ks 5..7: slots 10..15 = odd run + (2 ks + h - 9) * 16
            const unsigned wbase = (unsigned)q.off_w2 + (unsigned)(h * 32 + r) * 16u;    // lw2[(kh * 16 + 2 ks + h) * 32 + r]
            for (int tp = wave_u; kC2Nt * tp < ntile2; tp += 8) {
                unsigned abase[kC2Nt]; int trow[kC2Nt], moff[kC2Nt];
#pragma unroll
                for (int j = 0; j < kC2Nt; ++j) {
                    const int px = (kC2Nt * tp + j) * 32 + r;
                    if (c2_fast) {                                           // (uniform) whole-width band, one tile per wave: this lane's (row, column) were worked out once
                        abase[j] = c2s_abase;
                        trow[j] = s0 + 2 * (int)(c2s_rel >> 24);
                        moff[j] = px < npx2 ? m0 * 64 + (int)(c2s_rel & 0xFFFFFFu) : -1;
                    } else {
                        const int mm = min(px, npx2 - 1);                    // (lanes past the band's last pixel compute it again: never stored)
                        int yl2, x2;
                        divmod(mm, w2, inv_w2, yl2, x2);
                        abase[j] = (unsigned)q.off_tile + (unsigned)__mul24(x2, 48);   // even plane, pixel x2
                        trow[j] = s0 + 2 * yl2;                              // ring slot of conv1 row 2 yl2 + kh: trow + kh (mod NR)
                        moff[j] = px < npx2 ? ((m0 + __mul24(yl2, q.OW2) + x2) * 32 + 16 * h) * 2 : -1;   // this lane's 32 bytes of the pixel in conv2's activation
                    }
                }
                // the accumulators start at conv2's bias: register 4 qd + j = cout 8 qd + 4 h + j (read once per item, under the first fragments' latency)
                f32x16 acc2[kC2Nt];
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const float4 b = lb2[2 * qd + h];
#pragma unroll
                    for (int j = 0; j < kC2Nt; ++j) { acc2[j][4 * qd] = b.x; acc2[j][4 * qd + 1] = b.y; acc2[j][4 * qd + 2] = b.z; acc2[j][4 * qd + 3] = b.w; }
                }
                u4v fw[kC2Depth], fx[kC2Depth][kC2Nt];
                unsigned arow[kC2Nt], cur[kC2Nt];
                auto fetch = [&](int k, int d) {                             // k-step k = kernel row k / 8, window slots 2 (k % 8) + h (compile-time k after unrolling)
                    const int kh = k >> 3, ks = k & 7;
#pragma unroll
                    for (int j = 0; j < kC2Nt; ++j) {
                        if (ks == 0) {
                            int tr = trow[j] + kh;
                            tr = tr >= NR ? tr - NR : tr;
                            arow[j] = abase[j] + (unsigned)__mul24(tr, tile_pitch);
                            cur[j] = arow[j] + off_lo;
                        } else if (ks == 4) cur[j] = arow[j] + off_mid;
                        else if (ks == 5) cur[j] = arow[j] + off_hi;
                    }
                    const unsigned imm = ks == 4 ? 0u : 32u * ks;
                    fw[d] = *reinterpret_cast<const u4v*>(psmem + (wbase + (unsigned)(kh * 16 + 2 * ks) * 512u));
#pragma unroll
                    for (int j = 0; j < kC2Nt; ++j) fx[d][j] = *reinterpret_cast<const u4v*>(psmem + (cur[j] + imm));
                };
#pragma unroll
                for (int d = 0; d < kC2Depth; ++d) fetch(d, d);
#pragma unroll
                for (int k = 0; k < 40; ++k) {
                    const int d = k % kC2Depth;
                    const h16x8 w = __builtin_bit_cast(h16x8, fw[d]);
#pragma unroll
                    for (int j = 0; j < kC2Nt; ++j) {
#if TRS_C2_ABLATE == 2
                        asm volatile("" :: "v"(w), "v"(fx[d][j]));
#else
                        acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, __builtin_bit_cast(h16x8, fx[d][j]), acc2[j], 0, 0, 0);
#endif
                    }
                    if (k + kC2Depth < 40 && TRS_C2_ABLATE != 1) fetch(k + kC2Depth, d);
                    __builtin_amdgcn_sched_barrier(0);
                }
                BAND_STAMP(2);                                               // conv2: set-up and K loop (loaders: unpack + request)
#if TRS_C2_ABLATE == 3
                asm volatile("" :: "v"(acc2[0]));
                continue;
#endif
                // Epilogue without LDS: ReLU + fp16, the two lanes of a pixel swap half of their channel quads (v_permlane32_swap_b32) and each stores
                // 32 contiguous bytes (couts 16 h .. + 15).  (Until round 4 the tile went through a wave-private LDS stage — four broadcast bias reads and
                // the transpose, six dependent LDS round trips at ~200 clocks each while the other waves saturate the LDS: 1.4 k of a band's 6.5 k clocks
                // of phase 2 per tile, profiles/r04_pilot_head_stamps.txt.)
#pragma unroll
                for (int j = 0; j < kC2Nt; ++j) {
                    uint2 w[4];
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd)
                        w[qd] = q.c2.relu ? relu_pack4(acc2[j][4 * qd], acc2[j][4 * qd + 1], acc2[j][4 * qd + 2], acc2[j][4 * qd + 3])
                                          : make_uint2(pack_h16x2(acc2[j][4 * qd], acc2[j][4 * qd + 1]), pack_h16x2(acc2[j][4 * qd + 2], acc2[j][4 * qd + 3]));
                    u4v g0, g1;
                    quad_groups(w, g0, g1);
                    if (moff[j] >= 0) {
                        __builtin_amdgcn_raw_buffer_store_b128(g0, rout2, moff[j], 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(g1, rout2, moff[j] + 16, 0, 0);
                    }
                }
            }
#endif
        }
        BAND_STAMP(3);                                                      // conv2: epilogue and stores (loaders: all of phase 2)
        __syncthreads();                                                    // the tile is free, the next band image is complete
        BAND_STAMP(4);                                                      // wait at the barrier
        wt = nxt;
    }
#if TRS_BAND_STAMPS
    __syncthreads();
    if (blockIdx.x == 7 && lane == 0 && (wave == 0 || wave == 3 || wave == 8))
        printf("band head, workgroup 7, wave %d, %d items [clocks]: conv1 tiles %lld | barrier %lld | conv2 K loop %lld | conv2 stores (loaders: phase 2) %lld | barrier %lld\n",
               wave, total, st_[0], st_[1], st_[2], st_[3], st_[4]);
#endif
}

// dense2 -> dense3 -> output in fp32 (keras_train.py:161-168), then KerasPilot.step for CNN_2D_SPD_CTL (keras_pilot.py:78-95)
struct TailParams {
    const float* h1;           // dense1 output before ReLU, fp32: h1_slices slabs of [n][100] (split-K partial sums, added here in slice order)
    int h1_slices; size_t h1_stride;
    const float *w2, *b2, *w3, *b3, *w4, *b4;    // Keras layouts [IN][OUT]
    float* raw_out;            // [n][2] or nullptr
    const float* speed;        // 'gym/speed' or nullptr (no post-processing)
    const uint8_t* mode;       // 'usr/mode' per car (TRS_MODE_*) or nullptr: a car outside AI / AI_STEERING gets (0, 0, 0) (keras_pilot.py:139)
    float *steer, *thr, *brk;  // next controls
    int n, act;
    float threshold, rev_mult, brk_mult, smooth_thr;
    int use_break, smooth, direct;   // direct: ModelType.CNN_2D (outputs are steering and throttle)
};

// raw outputs -> KerasPilot.step's controls for one frame (keras_pilot.py:78-95; ModelType.CNN_2D: :56-62)
__device__ __forceinline__ void tail_finish(const TailParams& p, int i, float out0, float out1)
{
    if (p.raw_out) { p.raw_out[2 * i] = out0; p.raw_out[2 * i + 1] = out1; }
    if (!p.act) return;
    if (p.mode && p.mode[i] != TRS_MODE_AI && p.mode[i] != TRS_MODE_AI_STEERING) { p.steer[i] = 0.0f; p.thr[i] = 0.0f; p.brk[i] = 0.0f; return; }
    float steering = out0 < -1.0f ? -1.0f : (out0 > 1.0f ? 1.0f : out0);           // __cap (keras_pilot.py:142-145)
    const float predicted = out1 * 20.0f;                                           // :83
    const float real = p.speed[i];
    const float kHalfPi = 1.57079632679489661923f;
    float delta = predicted * p.threshold - real;                                   // calcThrottle (mapping.py:23-28)
    float throttle = p.rev_mult * atanf(delta * 2.0f) / kHalfPi;
    if (throttle > -0.2f && throttle < 0.0f) throttle = 0.0f;
    float breaking = 0.0f;
    if (p.direct) throttle = out1 < -1.0f ? -1.0f : (out1 > 1.0f ? 1.0f : out1);    // ModelType.CNN_2D: __cap of both outputs (:60)
    else if (p.use_break) {                                                         // keras_pilot.py:88-90, mapping.py:30-35
        throttle = (predicted - real > 0.0f) ? 1.0f : 0.0f;
        breaking = -1.0f * p.brk_mult * atanf(delta * 1.0f) / kHalfPi;
        if (breaking < 0.4f) breaking = 0.0f;
    }
    if (p.smooth) {                                                                 // keras_pilot.py:147-153
        if (steering > p.smooth_thr) steering = 1.0f;
        else if (steering < -p.smooth_thr) steering = -1.0f;
    }
    p.steer[i] = steering; p.thr[i] = throttle; p.brk[i] = breaking;
}

__global__ __launch_bounds__(256) void trs_pilot_tail_kernel(const TailParams p)
{
    // one wave per frame: lane o owns output neuron o of the current layer; activations travel through LDS
    __shared__ float s1[4][100], s2[4][50], s3[4][25], s4[4][2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wv;
    if (i >= p.n) return;
    // The three small layers are latency chains (100 + 50 + 25 dependent FMAs per lane); what made them slow was a global weight
    // load in front of every FMA.  All of a lane's weights — column `lane` of each matrix — are requested up front (175 independent
    // loads, coalesced across lanes) and land while the slab sums run; the FMAs then read registers and LDS only.
    float wc2[100], wc3[50], wc4[25];
    {
        const int o2 = min(lane, 49), o3 = min(lane, 24), o4 = min(lane, 1);
#pragma unroll
        for (int k = 0; k < 100; ++k) wc2[k] = p.w2[k * 50 + o2];
#pragma unroll
        for (int k = 0; k < 50; ++k) wc3[k] = p.w3[k * 25 + o3];
#pragma unroll
        for (int k = 0; k < 25; ++k) wc4[k] = p.w4[k * 2 + o4];
    }
    const float bias2 = p.b2[min(lane, 49)], bias3 = p.b3[min(lane, 24)], bias4 = p.b4[min(lane, 1)];
    {   // dense1's 100 outputs of this frame = the K-slice slabs added in slice order (fixed order: run-to-run identical).  A lane
        // owns outputs `lane` and `lane + 64`; the phase is pure load latency, so 16 slices x 2 outputs are requested together.
        const bool two = lane + 64 < 100;
        const float* src0 = p.h1 + (size_t)i * 100 + lane;
        const float* src1 = src0 + (two ? 64 : 0);
        float x0 = 0.0f, x1 = 0.0f;
        // groups of 16 slices, the last one padded: a slice past the end is read at the last slice's address and added as 0.0f (x + 0.0f == x for
        // every x that can reach the ReLU's test).  31 slices (240x320) are two load round trips; the ragged tail used to go 4 + 4 + 4 + 1 + 1 + 1
        // slices at a time: seven dependent round trips, the whole difference between this kernel's 6.0 us at 120x160 and 8.8 us at 240x320.
        auto add_groups = [&](auto gc) {
            constexpr int G = decltype(gc)::value;
            for (int sl = 0; sl < p.h1_slices; sl += G) {
                float v0[G], v1[G];
#pragma unroll
                for (int q = 0; q < G; ++q) {
                    const int sq = min(sl + q, p.h1_slices - 1);
                    v0[q] = src0[(size_t)sq * p.h1_stride]; v1[q] = src1[(size_t)sq * p.h1_stride];
                }
#pragma unroll
                for (int q = 0; q < G; ++q) {
                    const bool in = sl + q < p.h1_slices;
                    x0 += in ? v0[q] : 0.0f; x1 += in ? v1[q] : 0.0f;
                }
            }
        };
        if (p.h1_slices <= 8) add_groups(std::integral_constant<int, 8>{});      // (120x160: 8 slices - no padded loads)
        else add_groups(std::integral_constant<int, 16>{});
        s1[wv][lane] = x0 > 0.f ? x0 : 0.f;                                 // dense1's ReLU
        if (two) s1[wv][lane + 64] = x1 > 0.f ? x1 : 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 50) {
        float s = bias2;
#pragma unroll
        for (int k = 0; k < 100; ++k) s = fmaf(s1[wv][k], wc2[k], s);
        s2[wv][lane] = s > 0.f ? s : 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 25) {
        float s = bias3;
#pragma unroll
        for (int k = 0; k < 50; ++k) s = fmaf(s2[wv][k], wc3[k], s);
        s3[wv][lane] = s > 0.f ? s : 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 2) {
        float s = bias4;
#pragma unroll
        for (int k = 0; k < 25; ++k) s = fmaf(s3[wv][k], wc4[k], s);
        s4[wv][lane] = s;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane != 0) return;
    tail_finish(p, i, s4[wv][0], s4[wv][1]);
}

// ---- dense1 as its own kernel (round 2) -------------------------------------------------------------------------------------
// dense1 (4608 -> 100 at 120x160, 83,328 -> 100 at 240x320) is 0.8 % of the network's arithmetic but, as a 1x1 convolution on
// the chunked kernel, it and the tail were 23 of 267 us: 36 K slices of partial sums written and read back (19 MB against 9.4 MB
// of activations).  Here a workgroup takes 32 frames x one K slice (8 slices fill the chip at 1024 frames):
//   * the slice's activations go through LDS (coalesced 16-byte loads of each frame's contiguous run, chunks of 72 granules,
//     double buffered; row pitch 73 granules = conflict-free ds_read_b128 fragments);
//   * wave w owns channels 32 w .. 32 w + 31: all 36 k-steps of a chunk on one accumulator, its weight fragments ALL in flight (a
//     register ring of 36 granules straight from L2, each refilled for the next chunk right behind the MFMA that read it);
//   * workgroups are numbered so that one XCD works on one K slice: its 1/8 of the weights stays in that XCD's L2;
//   * the slice's partial sums go to slab [slice][frame][100]; the tail kernel adds the slabs in slice order.
// 15.5 -> 7.3 us (and the tail 7.5 -> 5.8 with 8 slabs instead of 36); 47 -> 36 us at 512 x 240x320.  Measured and NOT kept
// (profiles/r02_pilot_dense.txt): the tail in the same launch, run by the last K slice of a frame group to arrive — the hand-over
// between XCDs costs a write-through drain (2.2 us) and an sc1 read (4 us) and leaves 32 workgroups to do what 256 do in the
// tail kernel: 17 us in one launch against 7.3 + 5.8 in two.
struct DenseParams {
    const u4v* act;            // conv7's output: fp16 [n][G] granules (the NHWC flatten)
    const u4v* w;              // [G][128] granules
    const float* bias;         // [128]
    float* slab;               // [KS][n][100] fp32
    int n, G, gps, KS, groups;
};

#ifndef TRS_DENSE_ABLATE
#define TRS_DENSE_ABLATE 0   /* timing-only diagnostic builds (scripts/r04_dense_ablate.sh), never shipped */
#endif
constexpr int kDenseChunk = 72;            // granules per LDS chunk = 36 k-steps
constexpr int kDensePitch = 73;            // LDS row pitch in granules (odd: 16 lanes cover all 64 banks)
constexpr int kDenseSteps = kDenseChunk / 2;
constexpr int kDenseStage = 9;             // granules a thread stages per chunk and per 32 frames (32 x 72 / 256)
constexpr int kDenseLds = 2 * 32 * kDensePitch * 16;   // per 32 frames (NF = 2: twice that)

// NF = frame blocks of 32 per workgroup.  NF = 2 (round 4, long K: 240x320): every weight fragment feeds TWO MFMAs (frames 0..31 and 32..63), so the
// weight stream — per-lane 16-byte loads straight from L2, 1.18 MB per workgroup at 512 x 240x320, measured as 15 of the kernel's 37 us by a
// timing-only build (profiles/r04_pilot_dense.txt) — is halved for the same arithmetic; 150 KB of LDS, one workgroup per CU.
template <int NF>
__global__ __launch_bounds__(256) void trs_pilot_dense_kernel(const DenseParams p)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int group = idx % p.groups, slice = xcd + 8 * (idx / p.groups);
    if (slice >= p.KS) return;                                              // the grid is padded to whole rounds of 8 slices
    u4v* const st = reinterpret_cast<u4v*>(psmem);                          // [2][32 NF][73] granules
    const int gs = slice * p.gps, ge = min(p.G, gs + p.gps);
    const int f0 = group * 32 * NF;
    constexpr int kStage = kDenseStage * NF, kBuf = 32 * NF * kDensePitch;

    // staging: thread t moves granules e = t + 256 i (i < 9 NF) of a chunk: frame e / 72, granule e % 72
    int sbase[kStage], sdst[kStage];
#pragma unroll
    for (int i = 0; i < kStage; ++i) {
        const int e = tid + 256 * i, f = e / kDenseChunk, gi = e - f * kDenseChunk;
        sbase[i] = min(f0 + f, p.n - 1) * p.G;
        sdst[i] = f * kDensePitch + gi;
    }
    u4v pf[kStage];
    auto stage_load = [&](int c0) {
#pragma unroll
        for (int i = 0; i < kStage; ++i) {
            const int e = tid + 256 * i, f = e / kDenseChunk, gi = e - f * kDenseChunk;
#if TRS_DENSE_ABLATE == 2   /* timing-only: every chunk re-reads the slice's first chunk (cache hits instead of the HBM stream) */
            pf[i] = p.act[sbase[i] + min(gs + gi, p.G - 1)]; (void)c0;
#else
            pf[i] = p.act[sbase[i] + min(c0 + gi, p.G - 1)];
#endif
        }
    };
    auto stage_write = [&](int buf) {
        u4v* const sn = st + buf * kBuf;
#pragma unroll
        for (int i = 0; i < kStage; ++i) sn[sdst[i]] = pf[i];
    };
    // weight ring: granule 2 j + h of the chunk for k-step j, this wave's 32 channels
    u4v wr[kDenseSteps];
    const u4v* const wlane = p.w + wave * 32 + r;
    auto ring_load = [&](int j, int c0) { wr[j] = wlane[(size_t)min(c0 + 2 * j + h, p.G - 1) * 128]; };
    stage_load(gs);
#pragma unroll
    for (int j = 0; j < kDenseSteps; ++j) ring_load(j, gs);

    f32x16 acc[NF];
#pragma unroll
    for (int b = 0; b < NF; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = slice == 0 ? p.bias[wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * h] : 0.0f;
    stage_write(0);
    __syncthreads();

    auto chunk = [&](int c0, int buf, auto prefetch) {
        const u4v* const sb = st + buf * kBuf + r * kDensePitch + h;
#pragma unroll
        for (int j = 0; j < kDenseSteps; ++j) {
            const bool past = c0 + 2 * j >= ge;                             // ragged last chunk: a zero fragment instead of a branch
#pragma unroll
            for (int b = 0; b < NF; ++b) {
                u4v a = sb[b * 32 * kDensePitch + 2 * j];
                if (past) a = u4v{0u, 0u, 0u, 0u};
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, wr[j]), __builtin_bit_cast(h16x8, a), acc[b], 0, 0, 0);
            }
#if TRS_DENSE_ABLATE != 1   /* 1: timing-only, the weights of the first chunk serve every chunk (no weight stream) */
            if constexpr (decltype(prefetch)::value) ring_load(j, c0 + kDenseChunk);
#endif
        }
    };
    int c0 = gs, buf = 0;
    for (; c0 + kDenseChunk < ge; c0 += kDenseChunk, buf ^= 1) {            // every chunk but the last: the next one's operands are requested underneath
        stage_load(c0 + kDenseChunk);
        chunk(c0, buf, std::true_type{});
        stage_write(buf ^ 1);
        __syncthreads();
    }
    chunk(c0, buf, std::false_type{});

    {   // D[cout][frame]: lane = frame r, registers 4 q4 .. + 3 = channels 32 wave + 8 q4 + 4 h .. + 3: one 16-byte store each
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.slab, 0, (int)((size_t)p.KS * p.n * 400), 0x00020000);
#pragma unroll
        for (int b = 0; b < NF; ++b) {
            const int fr = f0 + 32 * b + r;
            const int row = ((slice * p.n + fr) * 100) * 4;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int c = wave * 32 + 8 * q4 + 4 * h;
                if (c < 100 && fr < p.n) {
                    const u4v v = {__float_as_uint(acc[b][4 * q4]), __float_as_uint(acc[b][4 * q4 + 1]), __float_as_uint(acc[b][4 * q4 + 2]), __float_as_uint(acc[b][4 * q4 + 3])};
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs, row + c * 4, 0, 0);
                }
            }
        }
    }
}

// The tail of the model types with extra inputs (components/keras_train.py): cnn_2d_speed_as_feature =
// Keras_2D_CNN.get_model(num_feature_vectors = 1) (:127-174: speed / 20 -> Dense 4 -> 8 -> 16, concatenated behind the flatten in
// front of dense1) and cnn_2d_full_house = Keras_2D_FULL_HOUSE.get_model (:184-245: 'loc/segment' -> 16 -> 32 -> 64 joins the
// flatten for the speed head dense1..3 -> output_speed; speed / 20 -> 16 -> 32 -> 64 joins both for the steering head
// dense4..6 -> out_steering; output = [steering, speed]).  The rows of dense1 / dense4 that multiply the flatten ran on the
// matrix cores (split-K slabs, added here in slice order); the rows of the small branches and everything behind are fp32, one
// wave per frame, a fixed summation order (k ascending) so that two runs agree bit for bit.
enum { XO_F1W, XO_F1B, XO_F2W, XO_F2B, XO_F3W, XO_F3B, XO_W1Y, XO_W2, XO_B2, XO_W3, XO_B3, XO_W4, XO_B4,
       XO_C1W, XO_C1B, XO_C2W, XO_C2B, XO_C3W, XO_C3B, XO_W4Y, XO_W5, XO_B5, XO_W6, XO_B6, XO_W7, XO_B7, XO_COUNT };

struct TailExParams {
    const float* h1; int h1_slices; size_t h1_stride;       // dense1: slabs of [n][100]
    const float* h4; int h4_slices; size_t h4_stride;       // dense4 (full house) or nullptr
    const float* blob; int xo[32];
    int arch, f1, f2, f3;                                    // widths of the small branches (4, 8, 16 or 16, 32, 64)
    const float* speed; const float* segment; const int32_t* seg_idx; int np;   // 'gym/speed'; 'loc/segment' given, or from the env's index
    const uint8_t* mode;
    float* raw_out; float *steer, *thr, *brk;
    int n, act;
    float threshold, rev_mult, brk_mult, smooth_thr;
    int use_break, smooth;
};

__device__ __forceinline__ void dense_small(const float* in, int nin, const float* w, const float* b, int nout, float* out, int lane, bool relu)
{   // out[o] = act(b[o] + sum_k in[k] * w[k][o]), lanes over o (nout <= 128).  One FMA chain per output, k ascending; the weights of 16 k
    // are requested together (a load in front of every FMA made this the slowest kernel of the full-house loop: one L2 round trip per k)
    for (int o = lane; o < nout; o += 64) {
        float s = b[o];
        int k = 0;
        for (; k + 16 <= nin; k += 16) {
            float wv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) wv[j] = w[(k + j) * nout + o];
#pragma unroll
            for (int j = 0; j < 16; ++j) s = fmaf(in[k + j], wv[j], s);
        }
        for (; k + 4 <= nin; k += 4) {
            float wv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wv[j] = w[(k + j) * nout + o];
#pragma unroll
            for (int j = 0; j < 4; ++j) s = fmaf(in[k + j], wv[j], s);
        }
        for (; k < nin; ++k) s = fmaf(in[k], w[k * nout + o], s);
        out[o] = relu ? (s > 0.f ? s : 0.f) : s;
    }
}

__global__ __launch_bounds__(256) void trs_pilot_tail_ex_kernel(const TailExParams p)
{
    __shared__ float sy[4][3][64], ss[4][3][64], sh[4][2][100], sa[4][2][50], sb[4][2][25], so[4][2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wv;
    if (i >= p.n) return;
    const float* B = p.blob;
    const float real = p.speed[i];
    // the small branches
    float fin = real / 20.0f;                                                      // keras_pilot.py:68,100: speed / 20
    if (p.arch == TRS_PILOT_FULL_HOUSE)                                            // 'loc/segment' (track_data_process.py:106-107: idx / len * 10, binary64)
        fin = p.segment ? p.segment[i] : (float)((double)p.seg_idx[i] / (double)p.np * 10.0);
    if (lane == 0) sy[wv][2][63] = fin;
    __builtin_amdgcn_wave_barrier();
    dense_small(&sy[wv][2][63], 1, B + p.xo[XO_F1W], B + p.xo[XO_F1B], p.f1, sy[wv][0], lane, true);
    __builtin_amdgcn_wave_barrier();
    dense_small(sy[wv][0], p.f1, B + p.xo[XO_F2W], B + p.xo[XO_F2B], p.f2, sy[wv][1], lane, true);
    __builtin_amdgcn_wave_barrier();
    dense_small(sy[wv][1], p.f2, B + p.xo[XO_F3W], B + p.xo[XO_F3B], p.f3, sy[wv][2], lane, true);
    if (p.arch == TRS_PILOT_FULL_HOUSE) {
        if (lane == 0) ss[wv][2][63] = real / 20.0f;
        __builtin_amdgcn_wave_barrier();
        dense_small(&ss[wv][2][63], 1, B + p.xo[XO_C1W], B + p.xo[XO_C1B], p.f1, ss[wv][0], lane, true);
        __builtin_amdgcn_wave_barrier();
        dense_small(ss[wv][0], p.f1, B + p.xo[XO_C2W], B + p.xo[XO_C2B], p.f2, ss[wv][1], lane, true);
        __builtin_amdgcn_wave_barrier();
        dense_small(ss[wv][1], p.f2, B + p.xo[XO_C3W], B + p.xo[XO_C3B], p.f3, ss[wv][2], lane, true);
    }
    __builtin_amdgcn_wave_barrier();
    // dense1 (and dense4): slabs in slice order, then the rows of the small branches, then ReLU
    for (int head = 0; head < (p.arch == TRS_PILOT_FULL_HOUSE ? 2 : 1); ++head) {
        const float* slab = head ? p.h4 : p.h1;
        const int slices = head ? p.h4_slices : p.h1_slices;
        const size_t stride = head ? p.h4_stride : p.h1_stride;
        const float* wy = B + p.xo[head ? XO_W4Y : XO_W1Y];
        for (int o = lane; o < 100; o += 64) {
            float x = 0.0f;
            {   // the K-slice slabs in slice order, eight loads in flight
                int sl = 0;
                for (; sl + 8 <= slices; sl += 8) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = slab[(size_t)(sl + j) * stride + (size_t)i * 100 + o];
#pragma unroll
                    for (int j = 0; j < 8; ++j) x += v[j];
                }
                for (; sl < slices; ++sl) x += slab[(size_t)sl * stride + (size_t)i * 100 + o];
            }
            for (int k = 0; k < p.f3; k += 4) {                             // f3 is 16 or 64
                float w4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) w4[j] = wy[(k + j) * 100 + o];
#pragma unroll
                for (int j = 0; j < 4; ++j) x = fmaf(sy[wv][2][k + j], w4[j], x);
            }
            if (head) for (int k = 0; k < p.f3; k += 4) {
                float w4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) w4[j] = wy[(p.f3 + k + j) * 100 + o];
#pragma unroll
                for (int j = 0; j < 4; ++j) x = fmaf(ss[wv][2][k + j], w4[j], x);
            }
            sh[wv][head][o] = x > 0.f ? x : 0.f;
        }
    }
    __builtin_amdgcn_wave_barrier();
    const int nz = p.arch == TRS_PILOT_FULL_HOUSE ? 1 : 2;                         // outputs of the first head
    dense_small(sh[wv][0], 100, B + p.xo[XO_W2], B + p.xo[XO_B2], 50, sa[wv][0], lane, true);
    if (p.arch == TRS_PILOT_FULL_HOUSE) dense_small(sh[wv][1], 100, B + p.xo[XO_W5], B + p.xo[XO_B5], 50, sa[wv][1], lane, true);
    __builtin_amdgcn_wave_barrier();
    dense_small(sa[wv][0], 50, B + p.xo[XO_W3], B + p.xo[XO_B3], 25, sb[wv][0], lane, true);
    if (p.arch == TRS_PILOT_FULL_HOUSE) dense_small(sa[wv][1], 50, B + p.xo[XO_W6], B + p.xo[XO_B6], 25, sb[wv][1], lane, true);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        auto lin = [&](const float* in, const float* w, const float* b, int nout, int o) {
            float s = b[o];
            for (int k = 0; k < 25; ++k) s = fmaf(in[k], w[k * nout + o], s);
            return s;
        };
        if (p.arch == TRS_PILOT_FULL_HOUSE) {                                      // [out_steering, out_speed] (keras_train.py:240)
            so[wv][0] = lin(sb[wv][1], B + p.xo[XO_W7], B + p.xo[XO_B7], 1, 0);
            so[wv][1] = lin(sb[wv][0], B + p.xo[XO_W4], B + p.xo[XO_B4], nz, 0);
        } else {
            so[wv][0] = lin(sb[wv][0], B + p.xo[XO_W4], B + p.xo[XO_B4], nz, 0);
            so[wv][1] = lin(sb[wv][0], B + p.xo[XO_W4], B + p.xo[XO_B4], nz, 1);
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane != 0) return;
    const float out[2] = {so[wv][0], so[wv][1]};
    if (p.raw_out) { p.raw_out[2 * i] = out[0]; p.raw_out[2 * i + 1] = out[1]; }
    if (!p.act) return;
    if (p.mode && p.mode[i] != TRS_MODE_AI && p.mode[i] != TRS_MODE_AI_STEERING) { p.steer[i] = 0.0f; p.thr[i] = 0.0f; p.brk[i] = 0.0f; return; }
    float steering = out[0] < -1.0f ? -1.0f : (out[0] > 1.0f ? 1.0f : out[0]);
    float throttle, breaking = 0.0f;
    if (p.arch == TRS_PILOT_SPD_FTR) {                                             // keras_pilot.py:67-76: both outputs capped, breaking 0
        throttle = out[1] < -1.0f ? -1.0f : (out[1] > 1.0f ? 1.0f : out[1]);
    } else {                                                                       // full house: the speed controller (:97-118)
        const float kHalfPi = 1.57079632679489661923f;
        const float predicted = out[1] * 20.0f;
        const float delta = predicted * p.threshold - real;
        throttle = p.rev_mult * atanf(delta * 2.0f) / kHalfPi;
        if (throttle > -0.2f && throttle < 0.0f) throttle = 0.0f;
        if (p.use_break) {
            throttle = (predicted - real > 0.0f) ? 1.0f : 0.0f;
            breaking = -1.0f * p.brk_mult * atanf(delta * 1.0f) / kHalfPi;
            if (breaking < 0.4f) breaking = 0.0f;
        }
    }
    if (p.smooth) {
        if (steering > p.smooth_thr) steering = 1.0f;
        else if (steering < -p.smooth_thr) steering = -1.0f;
    }
    p.steer[i] = steering; p.thr[i] = throttle; p.brk[i] = breaking;
}

// fp16 range check (trs_pilot_range_check): activations are stored as binary16 and SATURATE at 65504 (v_pk_min_i16 in every epilogue; the
// reference computes in fp32, components/keras_pilot.py:49-59).  A stored value of exactly 0x7BFF is a saturated one (a sum that lands on
// 65504 by itself is not a practical case): counted per layer into `per_layer` and into the handle's TRS_F_STATS slot.
__global__ __launch_bounds__(256) void trs_pilot_count_sat_kernel(const unsigned short* act, size_t n, unsigned long long* per_layer, unsigned long long* total)
{
    unsigned cnt = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) cnt += act[i] == 0x7BFFu ? 1u : 0u;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if ((threadIdx.x & 63) == 0 && cnt) { atomicAdd(per_layer, (unsigned long long)cnt); atomicAdd(total, (unsigned long long)cnt); }
}

__global__ void trs_zero_controls_kernel(float* a, float* b, float* c, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = 0.f; b[i] = 0.f; c[i] = 0.f; }
}

// ---------------------------------------------------------------------------------------------

#define HIPCHK(call)                                                                              \
    do {                                                                                          \
        hipError_t _e = (call);                                                                   \
        if (_e != hipSuccess) return trs_internal_fail(TRS_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(_e)); \
    } while (0)

struct ConvLayer {
    int KH, KW, S, CIN, COUT, COUT_PAD, IH, IW, OH, OW, G, G_pad;
    bool u8in, out_f32, relu;
    // conv1..7 on the single-layer kernels (trs_conv_u8_kernel, trs_conv_lt_kernel, trs_conv_span_kernel): the weights (or a 64-channel
    // slice) live in LDS, persistent workgroups
    int res_nb = 1, res_ysplit = 1, res_lds = 0, res_block = 512, res_wg_per_cu = 1;
    bool frame = false; int frame_f = 1, frame_lds = 0, frame_bands = 1, frame_ohb = 0;   // trs_conv_frame_kernel (3x3 stride-1 layers: F frames' input activations in LDS)
    bool frame5 = false; int frame5_lds = 0, frame5_bands = 1, frame5_ohb = 0;   // trs_conv_frame5_kernel (conv3: 5x5 stride 2 over 32 channels, one input frame / band per workgroup in LDS)
    bool res_span = false; int span_nl = 0, run_pad = 0;   // trs_conv_span_kernel (stride-2 5x5 layers: per-row input spans staged in LDS)
    u4v* w = nullptr; float* bias = nullptr; int* goff = nullptr;
};

struct PilotCtx {
    int n_cap = 0, H = 0, W = 0, cu_count = 256;
    ConvLayer L[9];                       // conv1..7 + dense1 (1x1 "conv" over frames) [+ dense4: the second head of cnn_2d_full_house]
    void* act[9] = {};                    // outputs of L[i] for n_cap frames (fp16; act[7], act[8] float)
    size_t act_elems[9] = {};             // per frame
    int arch = 0;                         // TRS_PILOT_SPD_CTL / CNN_2D share Keras_2D_CNN(2 outputs); TRS_PILOT_SPD_FTR (+1 feature vector); TRS_PILOT_FULL_HOUSE
    int n_layers = 8;
    float* xblob = nullptr;               // small fp32 weights of the extra dense branches (TailExParams offsets)
    int xo[32] = {};                      // offsets (floats) into xblob
    void* slab2 = nullptr; size_t slab2_bytes = 0; int last_slices2 = 1;   // dense4 partial sums
    float *w2 = nullptr, *b2 = nullptr, *w3 = nullptr, *b3 = nullptr, *w4 = nullptr, *b4 = nullptr;
    float* raw = nullptr;                 // [n_cap][2]
    uint8_t* tmp_frames = nullptr; size_t tmp_cap = 0;
    int last_n = 0, last_slices = 1;
    bool no_fuse = false;
    bool fuse12 = false; int fuse_r2 = 0, fuse_lds = 0; Fuse12Params fuse{};   // conv1 -> conv2 in one kernel, band form (conv1's activation stays in LDS)
    const uint8_t* last_frames = nullptr; bool act0_valid = false;           // conv1's activation is only materialised on demand (debug getter)
    void* slab = nullptr; size_t slab_bytes = 0;   // dense1 partial sums [slices][n][100] fp32
    int chain_first = -1, chain_lds = 0; ChainParams chain{};   // conv(chain_first + 1) .. conv7 in one launch (trs_conv_chain_kernel); -1: layer by layer
    bool chain_mid_valid = true;          // act[chain_first .. 5] hold the last pass (the chain never writes them; the debug getter runs the single layers on demand)
    u4v* w2_parity = nullptr;             // conv2's granules in the band kernel's order: per kernel row the even conv1 columns (kw 0, 2, 4), then the odd (1, 3)
    trs_pilot_tuning tun{};               // the kernel choices this context was loaded with (trs_pilot_set_tuning, else the defaults)
};

unsigned short host_f2h(float f)
{   // round to nearest even; weights beyond binary16's range saturate (|w| > 65504 does not occur in a trained network)
    const _Float16 h = (_Float16)std::min(std::max(f, -65504.0f), 65504.0f);
    unsigned short u; std::memcpy(&u, &h, 2);
    return u;
}
float host_h2f(unsigned short u) { _Float16 h; std::memcpy(&h, &u, 2); return (float)h; }

void free_ctx(PilotCtx* c)
{
    if (!c) return;
    for (auto& l : c->L) { (void)hipFree(l.w); (void)hipFree(l.bias); (void)hipFree(l.goff); }
    for (auto& a : c->act) (void)hipFree(a);
    (void)hipFree(c->w2); (void)hipFree(c->b2); (void)hipFree(c->w3); (void)hipFree(c->b3); (void)hipFree(c->w4); (void)hipFree(c->b4);
    (void)hipFree(c->raw); (void)hipFree(c->tmp_frames); (void)hipFree(c->slab); (void)hipFree(c->slab2); (void)hipFree(c->xblob); (void)hipFree(c->w2_parity);
    delete c;
}

template <typename T>
int upload(T** dst, const std::vector<T>& v)
{
    HIPCHK(hipMalloc((void**)dst, v.size() * sizeof(T)));
    HIPCHK(hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return TRS_OK;
}

int launch_conv(const ConvLayer& l, const void* in, size_t in_bytes, void* out, int n_img, hipStream_t s, int cu_count)
{
    ConvParams p{};
    p.in = in; p.w = l.w; p.bias = l.bias; p.goff = l.goff; p.out = out;
    if (in_bytes > 0x7FFFFFFFull) return trs_internal_fail(TRS_ERR_LIMIT, "activation larger than 2 GiB: lower the batch");
    p.in_bytes = (int)in_bytes;
    p.N = n_img; p.IH = l.IH; p.IW = l.IW; p.CIN = l.CIN; p.OH = l.OH; p.OW = l.OW; p.COUT = l.COUT; p.COUT_PAD = l.COUT_PAD; p.S = l.S;
    p.G = l.G; p.G_pad = l.G_pad; p.M = n_img * l.OH * l.OW;
    p.relu = l.relu; p.out_f32 = l.out_f32; p.in_px_bytes = l.u8in ? 3 : l.CIN * 2;
    p.oscale = l.u8in ? kConv1Scale : 1.0f;
    // outputs above 128 MB leave non-temporally (measured: conv1's 222 MB -> conv1 88 -> 81 us, conv2 85 -> 82; at 48 MB conv3 loses)
    p.nt_out = (!l.out_f32 && (size_t)p.M * l.COUT * 2 > ((size_t)128 << 20)) ? 1 : 0;
    p.KH = l.KH; p.KW = l.KW; p.run_pad = l.run_pad; p.cg = l.u8in ? 0 : l.CIN / 8; p.span_nl = l.span_nl;
    if (l.frame5) {                                                         // one persistent 8-wave workgroup per CU: a unit (frame or row band) computed from one LDS buffer, the next staged into the other
        Frame5Params q{};
        q.in = static_cast<const u4v*>(in); q.w = l.w; q.bias = l.bias; q.out = static_cast<unsigned short*>(out);
        q.N = n_img; q.IH = l.IH; q.IW = l.IW; q.OH = l.OH; q.OW = l.OW; q.F = 1; q.ev = (l.IW + 1) / 2; q.COUT = l.COUT;
        q.bands = l.frame5_bands; q.ohb = l.frame5_ohb; q.ihb = l.frame5_bands == 1 ? l.IH : 2 * l.frame5_ohb + 3;
        q.magic_ow = (unsigned)((0x100000000ull + (unsigned)l.OW - 1u) / (unsigned)l.OW);
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_conv_frame5_kernel<2, TRS_F5_NB, TRS_F5_R, 64 * (TRS_F5_COMPUTE + 4), TRS_F5_COMPUTE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL((trs_conv_frame5_kernel<2, TRS_F5_NB, TRS_F5_R, 64 * (TRS_F5_COMPUTE + 4), TRS_F5_COMPUTE>), dim3(std::min(n_img * q.bands, cu_count)), dim3(64 * (TRS_F5_COMPUTE + 4)), l.frame5_lds, s, q);
        HIPCHK(hipGetLastError());
        return TRS_OK;
    }
    if (l.frame) {
        FrameConvParams q{};
        q.in = static_cast<const u4v*>(in); q.w = l.w; q.bias = l.bias; q.out = static_cast<unsigned short*>(out);
        q.N = n_img; q.IH = l.IH; q.IW = l.IW; q.OH = l.OH; q.OW = l.OW; q.COUT = l.COUT; q.COUT_PAD = l.COUT_PAD; q.KH = l.KH; q.KW = l.KW;
        q.F = l.frame_f; q.cg = l.CIN / 8; q.cgs = q.cg == 8 ? 3 : 4; q.relu = l.relu;
        q.bands = l.frame_bands; q.ohb = l.frame_ohb; q.ihb = l.frame_ohb + l.KH - 1;
        { auto magic = [](int d) { return (unsigned)((0x100000000ull + (unsigned)d - 1u) / (unsigned)d); };
          q.magic_uout = magic(q.ohb * q.OW); q.magic_ow = magic(q.OW); q.magic_bands = q.bands == 1 ? 0u : magic(q.bands); }
        const int groups = (n_img * q.bands + q.F - 1) / q.F, grid = std::min(groups, cu_count);   // one persistent workgroup per CU
#define LAUNCH_FRAME(NT_, HALF_)                                                                                              \
    do {                                                                                                                      \
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_conv_frame_kernel<NT_, 2, HALF_, TRS_FRAME_R, 8, TRS_FRAME_LOADERS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((trs_conv_frame_kernel<NT_, 2, HALF_, TRS_FRAME_R, 8, TRS_FRAME_LOADERS>), dim3(grid), dim3(64 * (8 + TRS_FRAME_LOADERS)), l.frame_lds, s, q);   \
    } while (0)
        // (wave items of 2 x 32 pixels x 64 channels: the host cuts the frames into bands so that a group has about 8 of them — items of 3 tiles need
        // more than the 168 registers that 12 waves leave each)
        if (q.cg == 8) LAUNCH_FRAME(2, 4); else LAUNCH_FRAME(2, 8);
#undef LAUNCH_FRAME
        HIPCHK(hipGetLastError());
        return TRS_OK;
    }
    const int waves = l.res_block / 64, ntiles = (p.M + 31) / 32;
    const int grid_x = std::max(1, std::min((ntiles + waves - 1) / waves, cu_count * l.res_wg_per_cu));
#define LAUNCH_K(KERNEL)                                                                                                     \
    do {                                                                                                                     \
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, l.res_lds)); \
        hipLaunchKernelGGL((KERNEL), dim3(grid_x, l.res_ysplit), dim3(l.res_block), l.res_lds, s, p);                        \
    } while (0)
    if (l.u8in) LAUNCH_K(trs_conv_u8_kernel);                               // conv1 as its own layer
    else if (l.res_span && l.res_nb == 1) LAUNCH_K((trs_conv_span_kernel<1, false>));   // conv2 unfused (24 input channels: 6 granules per pixel pair, no swizzle)
    else if (l.res_span) LAUNCH_K((trs_conv_span_kernel<2, true>));         // conv3 where its frames do not fit LDS (240x320)
    else if (l.res_nb == 1) LAUNCH_K(trs_conv_lt_kernel<1>);                // the fallback of every other (layer, shape)
    else LAUNCH_K(trs_conv_lt_kernel<2>);
#undef LAUNCH_K
    HIPCHK(hipGetLastError());
    return TRS_OK;
}

// K slices of trs_pilot_dense_kernel: whole LDS chunks, as many slices as give every CU a workgroup (groups of 32 NF frames x slices).
// NF = 2 (64 frames per workgroup share every weight fragment) where the K dimension is long enough that every workgroup still gets two
// chunks or more (240x320: 8,816 granules; at 120x160 a slice is one chunk and the kernel is launch-bound: NF = 1).
void dense_plan(const ConvLayer& l, int n, int cu_count, const trs_pilot_tuning& T, int* gps, int* ks, int* nf_out = nullptr)
{
    const int G = l.G, chunks = (G + kDenseChunk - 1) / kDenseChunk;
    int nf = 1;
    {
        const int groups2 = (n + 63) / 64, want2 = std::max(1, cu_count / groups2);
        if (n >= 64 && (chunks + want2 - 1) / want2 >= 2) nf = 2;
        if (T.dense == 2) nf = 1;                                           // tuning: dense = 2 keeps 32 frames per workgroup (A/B)
    }
    const int groups = (n + 32 * nf - 1) / (32 * nf);
    int want = std::max(1, cu_count / groups);
    if (T.ksplit > 0) want = T.ksplit;
    const int cps = std::max(1, (chunks + want - 1) / want);
    *gps = cps * kDenseChunk;
    *ks = (G + *gps - 1) / *gps;
    if (nf_out) *nf_out = nf;
}

int launch_dense(PilotCtx* c, const ConvLayer& l, const void* in, int n, void* slab, hipStream_t s)
{
    DenseParams q{};
    q.act = static_cast<const u4v*>(in); q.w = l.w; q.bias = l.bias; q.slab = static_cast<float*>(slab);
    int nf = 1;
    q.n = n; q.G = l.G;
    dense_plan(l, n, c->cu_count, c->tun, &q.gps, &q.KS, &nf);
    q.groups = (n + 32 * nf - 1) / (32 * nf);
    const int grid = q.groups * ((q.KS + 7) / 8) * 8;
    if (nf == 2) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_pilot_dense_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kDenseLds));
        hipLaunchKernelGGL(trs_pilot_dense_kernel<2>, dim3(grid), dim3(256), 2 * kDenseLds, s, q);
    } else {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_pilot_dense_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, kDenseLds));
        hipLaunchKernelGGL(trs_pilot_dense_kernel<1>, dim3(grid), dim3(256), kDenseLds, s, q);
    }
    HIPCHK(hipGetLastError());
    return TRS_OK;
}

int forward(PilotCtx* c, const TrsEnvView& v, const uint8_t* d_frames, int n)
{
    const void* in = d_frames;
    size_t in_bytes = (size_t)n * c->H * c->W * 3;
    int first = 0;
    c->last_frames = d_frames; c->act0_valid = false;
    if (c->fuse12 && !c->no_fuse) {
        if (in_bytes > 0x7FFFFFFFull) return trs_internal_fail(TRS_ERR_LIMIT, "frames larger than 2 GiB: lower the batch");
        Fuse12Params q = c->fuse;
        q.frames = d_frames; q.frames_bytes = (int)in_bytes; q.N = n;
        q.c2.out = c->act[1]; q.c2.M = n * c->L[1].OH * c->L[1].OW;
        q.c2.nt_out = 0;
        int grid = std::max(1, std::min(n * q.bands * std::max(1, q.wsplit), c->cu_count));
        // rolling bands when there are enough (frame, part) streams for every CU (a small batch keeps one band per workgroup: more parallelism)
        q.roll = (c->tun.fuse_roll && n * std::max(1, q.wsplit) >= c->cu_count && q.bands > 1) ? 1 : 0;
        if (q.roll) grid = std::min(n * std::max(1, q.wsplit), c->cu_count);
        if (q.wsplit > 1) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_conv12_band_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, c->fuse_lds));
            hipLaunchKernelGGL(trs_conv12_band_kernel<true>, dim3(grid), dim3(1024), c->fuse_lds, v.stream, q);
        } else {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_conv12_band_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, c->fuse_lds));
            hipLaunchKernelGGL(trs_conv12_band_kernel<false>, dim3(grid), dim3(1024), c->fuse_lds, v.stream, q);
        }
        HIPCHK(hipGetLastError());
        in = c->act[1];
        in_bytes = (size_t)n * c->act_elems[1] * 2;
        first = 2;
    }
    c->chain_mid_valid = c->chain_first < 0;
    for (int i = first; i < 8; ++i) {
        if (i == c->chain_first) {                                          // conv(i + 1) .. conv7 in one launch, activations in LDS
            ChainParams q = c->chain;
            q.in = static_cast<const u4v*>(in); q.out = static_cast<unsigned short*>(c->act[6]); q.N = n;
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_conv_chain_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, c->chain_lds));
            hipLaunchKernelGGL(trs_conv_chain_kernel<512>, dim3((n + q.F - 1) / q.F), dim3(512), c->chain_lds, v.stream, q);
            HIPCHK(hipGetLastError());
            in = c->act[6];
            in_bytes = (size_t)n * c->act_elems[6] * 2;
            i = 6;
            continue;
        }
        void* out = c->act[i];
        if (i == 7) {                                                       // dense1: one fp32 slab per K slice, added in order by the tail kernel
            { int gps; dense_plan(c->L[7], n, c->cu_count, c->tun, &gps, &c->last_slices); }
            const size_t need = (size_t)c->last_slices * n * c->act_elems[7] * sizeof(float);
            if (c->slab_bytes < need) {
                HIPCHK(hipStreamSynchronize(v.stream));
                (void)hipFree(c->slab); c->slab = nullptr; c->slab_bytes = 0;
                HIPCHK(hipMalloc(&c->slab, need));
                c->slab_bytes = need;
            }
            out = c->slab;
        }
        int rc = i == 7 ? launch_dense(c, c->L[7], in, n, out, v.stream) : launch_conv(c->L[i], in, in_bytes, out, n, v.stream, c->cu_count);
        if (rc) return rc;
        if (i == 0) c->act0_valid = true;
        in = c->act[i];
        in_bytes = (size_t)n * c->act_elems[i] * (c->L[i].out_f32 ? 4 : 2);
    }
    if (c->arch == TRS_PILOT_FULL_HOUSE) {                                 // the steering head's dense4 reads conv7's output as well
        { int gps; dense_plan(c->L[8], n, c->cu_count, c->tun, &gps, &c->last_slices2); }
        const size_t need = (size_t)c->last_slices2 * n * c->act_elems[8] * sizeof(float);
        if (c->slab2_bytes < need) {
            HIPCHK(hipStreamSynchronize(v.stream));
            (void)hipFree(c->slab2); c->slab2 = nullptr; c->slab2_bytes = 0;
            HIPCHK(hipMalloc(&c->slab2, need));
            c->slab2_bytes = need;
        }
        int rc = launch_dense(c, c->L[8], c->act[6], n, c->slab2, v.stream);
        if (rc) return rc;
    }
    c->last_n = n;
    return TRS_OK;
}

struct ActIo { const float* speed; const float* segment; const uint8_t* mode; float *steer, *thr, *brk; };   // where KerasPilot.step's inputs / outputs live (device)

// the model type a caller asks for must be one the loaded weights can serve
int check_model_type(const PilotCtx* c, const trs_pilot_config* cfg)
{
    const int mt = cfg->model_type;
    const bool ok = c->arch == TRS_PILOT_SPD_CTL ? (mt == TRS_PILOT_SPD_CTL || mt == TRS_PILOT_CNN_2D) : mt == c->arch;
    if (!ok) return trs_internal_fail(TRS_ERR_ARG, "trs_pilot_config.model_type does not match the loaded weights (22 arrays: cnn_2d_speed_control / cnn_2d; 28: cnn_2d_speed_as_feature; 42: cnn_2d_full_house)");
    return TRS_OK;
}

TailParams make_tail(const PilotCtx* c, const TrsEnvView& v, int n, float* raw_out, const trs_pilot_config* cfg, bool act, const ActIo* io)
{
    TailParams t{};
    t.w2 = c->w2; t.b2 = c->b2; t.w3 = c->w3; t.b3 = c->b3; t.w4 = c->w4; t.b4 = c->b4;
    t.raw_out = raw_out; t.n = n; t.act = act ? 1 : 0;
    if (act) {
        t.speed = v.speed; t.steer = v.ctl_steer; t.thr = v.ctl_thr; t.brk = v.ctl_brk;
        if (io) { t.speed = io->speed ? io->speed : v.speed; t.mode = io->mode; t.steer = io->steer; t.thr = io->thr; t.brk = io->brk; }
        t.threshold = cfg->spd_ctl_threshold; t.rev_mult = cfg->spd_ctl_reverse_multiplier; t.brk_mult = cfg->spd_ctl_break_multiplier;
        t.use_break = cfg->spd_ctl_break; t.smooth = cfg->smooth_steering_enabled; t.smooth_thr = cfg->smooth_steering_threshold;
        t.direct = cfg->model_type == TRS_PILOT_CNN_2D;
    }
    return t;
}

int run_tail(PilotCtx* c, const TrsEnvView& v, int n, float* raw_out, const trs_pilot_config* cfg, bool act, const ActIo* io = nullptr)
{
    if (c->arch != TRS_PILOT_SPD_CTL) {
        TailExParams t{};
        t.h1 = static_cast<const float*>(c->slab); t.h1_slices = c->last_slices; t.h1_stride = (size_t)n * c->act_elems[7];
        t.h4 = static_cast<const float*>(c->slab2); t.h4_slices = c->last_slices2; t.h4_stride = (size_t)n * c->act_elems[8];
        t.blob = c->xblob; std::memcpy(t.xo, c->xo, sizeof t.xo);
        t.arch = c->arch; t.f1 = c->arch == TRS_PILOT_SPD_FTR ? 4 : 16; t.f2 = 2 * t.f1; t.f3 = 4 * t.f1;
        t.speed = (io && io->speed) ? io->speed : v.speed; t.segment = io ? io->segment : nullptr; t.seg_idx = v.seg_idx; t.np = v.n_points > 0 ? v.n_points : 1;
        t.raw_out = raw_out; t.n = n; t.act = act ? 1 : 0;
        if (act) {
            t.steer = v.ctl_steer; t.thr = v.ctl_thr; t.brk = v.ctl_brk;
            if (io) { t.mode = io->mode; t.steer = io->steer; t.thr = io->thr; t.brk = io->brk; }
            t.threshold = cfg->spd_ctl_threshold; t.rev_mult = cfg->spd_ctl_reverse_multiplier; t.brk_mult = cfg->spd_ctl_break_multiplier;
            t.use_break = cfg->spd_ctl_break; t.smooth = cfg->smooth_steering_enabled; t.smooth_thr = cfg->smooth_steering_threshold;
        }
        hipLaunchKernelGGL(trs_pilot_tail_ex_kernel, dim3((n + 3) / 4), dim3(256), 0, v.stream, t);
        HIPCHK(hipGetLastError());
        return TRS_OK;
    }
    TailParams t = make_tail(c, v, n, raw_out, cfg, act, io);
    t.h1 = static_cast<const float*>(c->slab); t.h1_slices = c->last_slices; t.h1_stride = (size_t)n * c->act_elems[7];
    hipLaunchKernelGGL(trs_pilot_tail_kernel, dim3((n + 3) / 4), dim3(256), 0, v.stream, t);
    HIPCHK(hipGetLastError());
    return TRS_OK;
}

// the whole pilot: convolutions, dense1 and the tail
int forward_and_tail(PilotCtx* c, const TrsEnvView& v, const uint8_t* d_frames, int n, float* raw_out, const trs_pilot_config* cfg, bool act, const ActIo* io = nullptr)
{
    int rc = forward(c, v, d_frames, n);
    if (rc) return rc;
    return run_tail(c, v, n, raw_out, cfg, act, io);
}

}  // namespace

void trs_pilot_free(void* ctx) { free_ctx(static_cast<PilotCtx*>(ctx)); }

TRS_EXPORT void trs_default_pilot_config(trs_pilot_config* c)
{
    if (!c) return;
    std::memset(c, 0, sizeof *c);
    c->struct_size = (uint32_t)sizeof *c;
    c->spd_ctl_threshold = 1.1f; c->spd_ctl_break = 0; c->spd_ctl_reverse_multiplier = 1.0f; c->spd_ctl_break_multiplier = 1.0f;
    c->smooth_steering_enabled = 0; c->smooth_steering_threshold = 0.9f;
}

TRS_EXPORT void trs_default_pilot_tuning(trs_pilot_tuning* t)
{
    if (!t) return;
    std::memset(t, 0, sizeof *t);
    t->struct_size = (uint32_t)sizeof *t;
    t->fuse_band_r2 = 6; t->fuse_wsplit_max = 4; t->span_layers_mask = 0x6;
    t->fuse_roll = 1; t->frame5 = 1; t->frame_layers_mask = 0x78; t->chain_layers = 4; t->dense = 1;
}

TRS_EXPORT int trs_pilot_set_tuning(trs_env* e, const trs_pilot_tuning* t)
{
    if (!e) return trs_internal_fail(TRS_ERR_ARG, "null handle");
    if (t && t->struct_size != sizeof(trs_pilot_tuning)) return trs_internal_fail(TRS_ERR_ARG, "trs_pilot_tuning.struct_size does not match this library (start from trs_default_pilot_tuning)");
    trs_internal_set_pilot_tuning(e, t);
    return TRS_OK;
}

TRS_EXPORT int trs_pilot_load(trs_env* e, const float* const* arr, int n_arrays)
{
    TrsEnvView v;
    if (!trs_internal_view(e, &v)) return trs_internal_fail(TRS_ERR_ARG, "null handle");
    if (!arr || (n_arrays != 2 * kLayers && n_arrays != 28 && n_arrays != 42))
        return trs_internal_fail(TRS_ERR_ARG, "expected 22 arrays: kernel and bias of conv1..conv7, dense1..dense3, output_layer (28 with feature1..3 for cnn_2d_speed_as_feature; "
                                              "42 for cnn_2d_full_house: ..., output_speed, feature1..3, current_spd_1..3, dense4..6, out_steering)");
    for (int i = 0; i < n_arrays; ++i) if (!arr[i]) return trs_internal_fail(TRS_ERR_ARG, "null weight array");
    HIPCHK(hipSetDevice(v.device));
    void** slot = trs_internal_pilot_slot(e);
    if (*slot) { HIPCHK(hipStreamSynchronize(v.stream)); free_ctx(static_cast<PilotCtx*>(*slot)); *slot = nullptr; }
    // the context under construction is freed on EVERY early return below (HIPCHK included); released into the handle at the end
    std::unique_ptr<PilotCtx, void (*)(PilotCtx*)> guard(new PilotCtx(), free_ctx);
    PilotCtx* const c = guard.get();
    c->n_cap = v.n; c->H = v.H; c->W = v.W;
    trs_default_pilot_tuning(&c->tun);
    if (const trs_pilot_tuning* user = trs_internal_pilot_tuning(e)) c->tun = *user;
    const trs_pilot_tuning& T = c->tun;
    c->arch = n_arrays == 28 ? TRS_PILOT_SPD_FTR : (n_arrays == 42 ? TRS_PILOT_FULL_HOUSE : TRS_PILOT_SPD_CTL);
    c->n_layers = c->arch == TRS_PILOT_FULL_HOUSE ? 9 : 8;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, v.device) == hipSuccess && prop.multiProcessorCount > 0) c->cu_count = prop.multiProcessorCount; }
    static const int spec[7][4] = {{5, 2, 3, 24}, {5, 2, 24, 32}, {5, 2, 32, 64}, {3, 1, 64, 64}, {3, 1, 64, 64}, {3, 1, 64, 128}, {3, 1, 128, 128}};
    int ih = v.H, iw = v.W;
    for (int i = 0; i < c->n_layers; ++i) {
        ConvLayer& l = c->L[i];
        if (i < 7) { l.KH = l.KW = spec[i][0]; l.S = spec[i][1]; l.CIN = spec[i][2]; l.COUT = spec[i][3]; l.IH = ih; l.IW = iw; }
        else { l.KH = l.KW = 1; l.S = 1; l.CIN = ih * iw * 128; l.COUT = 100; l.IH = 1; l.IW = 1; }     // dense1 (and dense4) over the NHWC flatten:
                                                                                                          // the kernel's first CIN rows; the rows of the small branches go to the tail
        l.OH = (l.IH - l.KH) / l.S + 1; l.OW = (l.IW - l.KW) / l.S + 1;
        if (l.OH < 1 || l.OW < 1) { return trs_internal_fail(TRS_ERR_LIMIT, "image too small for Keras_2D_CNN"); }
        l.COUT_PAD = (l.COUT + 31) / 32 * 32;
        l.u8in = (i == 0); l.relu = true; l.out_f32 = (i >= 7);
        // granules: a kernel row is one contiguous run of KW * CIN / 8 granules in NHWC; runs are padded to whole trips of 4
        // (zero weights) so that a trip is always 64 contiguous bytes (trs_conv_lt_kernel); dense1 is one long run
        const int run = l.u8in ? 2 : l.KW * l.CIN / 8, run_pad = l.u8in ? 2 : (run + 3) & ~3;
        l.G = l.KH * run_pad;
        l.G_pad = (l.G + 3) & ~3;
        if (i >= 7 && (l.COUT_PAD != 128 || l.G % 16 != 0)) return trs_internal_fail(TRS_ERR_LIMIT, "dense1's shape does not suit trs_pilot_dense_kernel");   // (never for Keras_2D_CNN: 100 outputs, 16 granules per pixel of conv7's output)
        if (i < 7) {       // conv layers on their single-layer kernels: resident weights, at most 64 output channels per slice (NB <= 2 keeps 16 waves per CU in registers)
            l.res_nb = std::min(2, l.COUT_PAD / 32);
            l.res_ysplit = l.COUT_PAD / (32 * l.res_nb);
            l.run_pad = run_pad;
            // stride-2 layers with wide kernels re-fetch every byte ~2.5x through overlapping windows: span staging instead
            const int cgr = l.CIN / 8, pix_gran = l.S * cgr;
            const int nseg_max = 30 / l.OW + 2;
            const int span_mask = T.span_layers_mask;                         // bit i = conv(i+1) uses the span kernel: conv2 and conv3 (0x6)
            l.res_span = !l.u8in && l.S == 2 && l.KW >= 5 && nseg_max <= 4 && ((span_mask >> i) & 1) && l.COUT % 32 == 0;   // (whole 32-channel blocks: its epilogue stores 16 channels per lane)
            if (l.res_span) {
                l.span_nl = ((32 - nseg_max) * pix_gran + nseg_max * run_pad + 63) / 64;
                if (l.span_nl > 5) l.res_span = false;
            }
            const int stage_per_wave = l.res_span ? l.span_nl * 1024 : 2048;  // input transpose / span stage; output transpose (all kernels)
            auto lds_for = [&](int nb, int waves) { return l.G_pad * nb * 32 * 16 + ((l.G_pad * 4 + 15) & ~15) + nb * 32 * 4 + waves * stage_per_wave; };
            // conv7's 64-channel slices at 7 waves beat 32-channel slices at 16 (240x320: 199 -> 143 us; the pixels are read twice instead of four times)
            if (lds_for(l.res_nb, 7) > 160 * 1024) { l.res_nb = 1; l.res_ysplit = l.COUT_PAD / 32; }    // 32-channel slices
            if (lds_for(l.res_nb, 4) > 160 * 1024) return trs_internal_fail(TRS_ERR_LIMIT, "a convolution's weight slice does not fit LDS");   // (never for Keras_2D_CNN)
            // workgroups per CU and waves per workgroup: about 16 waves per CU when LDS allows
            l.res_wg_per_cu = 1;
            for (int wg = 4; wg >= 1; --wg) {
                const int waves = std::max(4, 16 / wg);
                if (wg * (lds_for(l.res_nb, waves) + 512) <= 160 * 1024) { l.res_wg_per_cu = wg; break; }
            }
            int waves = std::max(4, 16 / l.res_wg_per_cu);
            if (l.res_span) waves = std::min(waves, 12);                      // trs_conv_span_kernel is built for <= 768 threads
            while (waves > 4 && l.res_wg_per_cu * (lds_for(l.res_nb, waves) + 512) > 160 * 1024) --waves;
            l.res_block = 64 * waves;
            l.res_lds = lds_for(l.res_nb, waves);
        }
        if (i >= 3 && i < 7) {   // conv4..7: frames in LDS when they fit (240x320: conv7's 167 KB frame does not: the quad-load kernel stays)
            const int mask = T.frame_layers_mask;                             // bit i = conv(i+1) (0x78)
            const int cg = l.CIN / 8;
            const bool shape_ok = l.S == 1 && l.KH == 3 && l.KW == 3 && (cg == 8 || cg == 16) && l.COUT_PAD % 64 == 0 && l.COUT == l.COUT_PAD && run_pad == l.KW * cg;
            if (((mask >> i) & 1) && shape_ok) {
                // Two LDS buffers (the group being computed and the next one), F units each, a unit = a frame or one of `bands` row bands of it (240x320:
                // conv4 128 KB, conv6 97 KB, conv7 167 KB per frame).  Among the (bands, F) that fit, take the one with the fewest MFMA rounds: a group is
                // ceil(tiles / 2) x (COUT / 64) wave items for 8 compute waves; bands re-stage KH - 1 rows each (a small penalty), every CU should get
                // at least two groups (one to compute, one on its way).
                int bands = 1, f = 1; double best = 1e30;
                for (int bnd = 1; bnd <= l.OH && bnd <= 8; ++bnd) {
                    const int ohb_ = (l.OH + bnd - 1) / bnd;
                    const size_t unit_ = (size_t)(ohb_ + l.KH - 1) * l.IW * l.CIN * 2;
                    for (int f_ = 1; f_ <= 8; ++f_) {
                        if (2 * unit_ * f_ + l.COUT_PAD * 4 > 158 * 1024) break;
                        const long groups = ((long)c->n_cap * bnd + f_ - 1) / f_;
                        if (f_ > 1 && groups < 2 * c->cu_count) break;
                        const int tiles = (f_ * ohb_ * l.OW + 31) / 32, items = ((tiles + 1) / 2) * (l.COUT_PAD / 64);
                        const double rounds = (double)((items + 7) / 8), per_cu = std::ceil((double)groups / c->cu_count);
                        const double cost = per_cu * (rounds + 0.15) * (1.0 + 0.1 * (double)(bnd - 1) * (l.KH - 1) / l.OH);   // + a barrier and a hand-over per group
                        if (cost < best - 1e-9) { best = cost; bands = bnd; f = f_; }
                    }
                }
                const int ohb = (l.OH + bands - 1) / bands;
                const int ihb = ohb + l.KH - 1;
                const size_t unit_bytes = (size_t)ihb * l.IW * l.CIN * 2;
                l.frame = best < 1e29;
                l.frame_f = f; l.frame_bands = bands; l.frame_ohb = ohb; l.frame_lds = (int)(2 * f * unit_bytes) + l.COUT_PAD * 4;
            }
        }
        if (i == 2) {            // conv3: frames in LDS when two fit (120x160: 2 x 64 KB); TRS_PILOT_FRAME5 = 0: the span kernel
            const int on = T.frame5;
            const bool shape_ok = l.KH == 5 && l.KW == 5 && l.S == 2 && l.CIN == 32 && l.COUT == 64 && l.COUT_PAD == 64 && run_pad == 20 && l.G_pad == 100;
            // a frame larger than ~78 KB is cut into row bands (240x320: 57 x 77 x 32 = 281 KB -> 5 bands of 6 output rows = 15 input rows, 74 KB:
            // two workgroups of one band per CU)
            int bands = 1;
            while (bands < l.OH && (size_t)(bands == 1 ? l.IH : 2 * ((l.OH + bands - 1) / bands) + 3) * l.IW * 64 > 78 * 1024) ++bands;
            const int ohb = (l.OH + bands - 1) / bands, ihb = bands == 1 ? l.IH : 2 * ohb + 3;
            const size_t unit = (size_t)ihb * l.IW * 64;
            // (round 3: one unit per 4-wave workgroup, two workgroups per CU; round 4: one persistent double-buffered workgroup per CU — see the kernel)
            // (row bands at 240x320: 89 us against the span kernel's 86 in round 3, one unit per workgroup; 72.0 against 83.1 on the persistent double-buffered
            // workgroup of round 4: bands by default where a frame does not fit)
            if (on && shape_ok && 2 * unit + 256 <= 158 * 1024) {           // two buffers: the unit being computed and the next one
                l.frame5 = true; l.frame5_lds = (int)(2 * unit) + 64 * 4; l.frame5_bands = bands; l.frame5_ohb = ohb;
            }
        }
        // ---- pack the kernel into granules [g][cout_pad][8] of fp16 and the per-granule input offsets ----
        const int ai = i == 8 ? 34 : 2 * i;                               // dense4 of the full-house model
        const float* K = arr[ai];
        const float* B = arr[ai + 1];
        std::vector<unsigned short> wp((size_t)l.G_pad * l.COUT_PAD * 8, 0);
        std::vector<int> goff(l.G_pad, 0);
        for (int g = 0; g < l.G; ++g) {
            if (l.u8in) {
                const int kh = g >> 1, half = g & 1;
                goff[g] = kh * l.IW * 3 + 8 * half;
                for (int j = 0; j < 8; ++j) {
                    const int f = 8 * half + j;                       // byte f of the 16-byte row window = (kw, c), 15 is padding
                    if (f >= 15) continue;
                    const int kw = f / 3, ch = f % 3;
                    for (int co = 0; co < l.COUT; ++co)
                        wp[((size_t)g * l.COUT_PAD + co) * 8 + j] = host_f2h(K[((kh * l.KW + kw) * l.CIN + ch) * l.COUT + co] * (256.0f / 255.0f));   // x 2^-8 in the epilogue (kConv1Scale)
                }
            } else {
                const int kh = g / run_pad, gi = g % run_pad;                 // granule gi of kernel row kh
                goff[g] = (kh * l.IW * l.CIN + gi * 8) * 2;
                if (gi >= run) continue;                                      // run padding: next pixel's bytes x zero weights
                const int c8n = l.CIN / 8, kw = gi / c8n, c8 = gi % c8n;
                for (int j = 0; j < 8; ++j)
                    for (int co = 0; co < l.COUT; ++co)
                        wp[((size_t)g * l.COUT_PAD + co) * 8 + j] = host_f2h(K[((kh * l.KW + kw) * l.CIN + c8 * 8 + j) * l.COUT + co]);
            }
        }
        for (int g = l.G; g < l.G_pad; ++g) goff[g] = goff[l.G - 1];       // padding granule: valid address, zero weights
        std::vector<float> bias(l.COUT_PAD, 0.0f);
        for (int co = 0; co < l.COUT; ++co) bias[co] = B[co];
        std::vector<u4v> wv(wp.size() / 8);
        std::memcpy(wv.data(), wp.data(), wp.size() * 2);
        int rc = upload(&l.w, wv); if (!rc) rc = upload(&l.bias, bias); if (!rc) rc = upload(&l.goff, goff);
        if (rc) return rc;
        c->act_elems[i] = (size_t)l.OH * l.OW * l.COUT;
        HIPCHK(hipMalloc(&c->act[i], (size_t)c->n_cap * c->act_elems[i] * (l.out_f32 ? 4 : 2) + 64));
        if (i < 7) { ih = l.OH; iw = l.OW; }
    }
    auto up = [&](float** dst, const float* src, size_t n) -> int { std::vector<float> t(src, src + n); return upload(dst, t); };
    const int nz = c->arch == TRS_PILOT_FULL_HOUSE ? 1 : 2;                // outputs of the first head
    int rc = up(&c->w2, arr[16], 100 * 50); if (!rc) rc = up(&c->b2, arr[17], 50);
    if (!rc) rc = up(&c->w3, arr[18], 50 * 25); if (!rc) rc = up(&c->b3, arr[19], 25);
    if (!rc) rc = up(&c->w4, arr[20], 25 * nz); if (!rc) rc = up(&c->b4, arr[21], nz);
    if (rc) return rc;
    if (c->arch != TRS_PILOT_SPD_CTL) {        // the small fp32 branches of the tail: one blob, offsets by XO_*
        const int F = (int)c->L[7].CIN;
        const int f1 = c->arch == TRS_PILOT_SPD_FTR ? 4 : 16, f2 = 2 * f1, f3 = 4 * f1;
        std::vector<float> blob;
        auto put = [&](int slot, const float* src, size_t n) { c->xo[slot] = (int)blob.size(); blob.insert(blob.end(), src, src + n); };
        put(XO_F1W, arr[22], f1); put(XO_F1B, arr[23], f1); put(XO_F2W, arr[24], (size_t)f1 * f2); put(XO_F2B, arr[25], f2);
        put(XO_F3W, arr[26], (size_t)f2 * f3); put(XO_F3B, arr[27], f3);
        put(XO_W1Y, arr[14] + (size_t)F * 100, (size_t)f3 * 100);            // dense1's rows behind the flatten
        put(XO_W2, arr[16], 100 * 50); put(XO_B2, arr[17], 50); put(XO_W3, arr[18], 50 * 25); put(XO_B3, arr[19], 25);
        put(XO_W4, arr[20], 25 * nz); put(XO_B4, arr[21], nz);
        if (c->arch == TRS_PILOT_FULL_HOUSE) {
            put(XO_C1W, arr[28], f1); put(XO_C1B, arr[29], f1); put(XO_C2W, arr[30], (size_t)f1 * f2); put(XO_C2B, arr[31], f2);
            put(XO_C3W, arr[32], (size_t)f2 * f3); put(XO_C3B, arr[33], f3);
            put(XO_W4Y, arr[34] + (size_t)F * 100, (size_t)2 * f3 * 100);    // dense4's rows for [feature3 | current_spd_3]
            put(XO_W5, arr[36], 100 * 50); put(XO_B5, arr[37], 50); put(XO_W6, arr[38], 50 * 25); put(XO_B6, arr[39], 25);
            put(XO_W7, arr[40], 25); put(XO_B7, arr[41], 1);
        }
        rc = upload(&c->xblob, blob);
        if (rc) return rc;
    }
    {   // conv1 -> conv2 fusion: needs the 5x5/2 + 5x5/2 head of Keras_2D_CNN and an LDS tile of 2 R2 + 3 conv1 rows
        const ConvLayer& l0 = c->L[0]; const ConvLayer& l1 = c->L[1];
        Fuse12Params& q = c->fuse;
        q = Fuse12Params{};
        q.w1 = l0.w; q.b1 = l0.bias; q.w2 = l1.w;
        q.c2 = ConvParams{};
        q.c2.bias = l1.bias; q.c2.COUT = l1.COUT; q.c2.COUT_PAD = l1.COUT_PAD; q.c2.relu = 1; q.c2.oscale = 1.0f;
        q.IH = l0.IH; q.IW = l0.IW; q.OH1 = l0.OH; q.OW1 = l0.OW; q.OH2 = l1.OH; q.OW2 = l1.OW;
        {   // can conv1 leave binary16's range?  Its inputs are pixels / 256 <= 1, so |output| <= |bias| + sum |w| (x 256 / 255 and the binary16 rounding of the
            // weights: 1.01 covers both).  Below 65504 for every output channel the fused head's conv1 epilogue needs no saturation step.
            const float* K0 = arr[0]; const float* B0 = arr[1];
            double worst = 0.0;
            for (int co = 0; co < l0.COUT; ++co) {
                double sum = std::fabs((double)B0[co]);
                for (int k = 0; k < l0.KH * l0.KW * l0.CIN; ++k) sum += std::fabs((double)K0[(size_t)k * l0.COUT + co]) * 1.01;
                worst = std::max(worst, sum);
            }
            q.c1_bounded = (worst == worst && worst < 60000.0) ? 1 : 0;      // (NaN weights: not bounded)
        }
        const bool shape_ok = l0.G_pad == 12 && l0.COUT_PAD == 32 && l1.G_pad == 80 && l1.COUT_PAD == 32 && l1.COUT == 32 && l1.CIN == 24 && l1.S == 2 && l1.KH == 5;
        const bool band_ok = l0.OW >= 32 && (2 * 8 + 3) * l0.OW < 65536;   // the band kernels split a tile's first pixel on the scalar unit and let a lane wrap once
        c->fuse12 = false; c->no_fuse = T.no_fuse != 0;
        // band form (conv1's input staged once per band as a fp16 image): tile + band image
        // (measured, 1024 frames of 120x160: R2 = 7 / 6 / 5 -> 129 / 114 / 125 us against 131 for the direct form; 512 frames of
        // 240x320, where only R2 = 2 fits: 290 against 272 - bands thinner than 4 rows recompute too much of conv1)
        const int band_r2 = T.fuse_band_r2;                               // 6; 0 = use the direct form
        for (int r2 = std::min(band_r2, l1.OH); shape_ok && band_ok && r2 >= std::min(4, l1.OH); --r2) {
            const int rows_in = 2 * (2 * r2 + 3) + 3, row_in = l0.IW * 3;
            if (row_in % 16 != 0 || rows_in * row_in > kBandPf * 512 * 16) continue;
            int off = 12 * 32 * 16;
            q.off_w2 = off; off += 80 * 32 * 16;
            q.off_b = off; off += 16 * 16;
            q.off_tile = off; q.tile_bytes = (((2 * r2 + 3) * 2 * ((l0.OW + 1) / 2) * 48 + 128) + 15) & ~15; off += q.tile_bytes;   // two column-parity planes per row
            q.off_band = off; q.band_bytes = rows_in * row_in * 2 + 64; off += q.band_bytes;
            if (off <= 160 * 1024) { c->fuse12 = true; c->fuse_r2 = r2; c->fuse_lds = off; q.R2 = r2; q.bands = (l1.OH + r2 - 1) / r2; q.wsplit = 1; break; }
        }
        // the band cut in width (240x320: a whole-width band does not fit): parts of w2p conv2 columns, each with its own conv1 tile and
        // staged frame-row segments (a part re-stages 2 x 3 + 3 input columns and recomputes 3 conv1 columns of its neighbour)
        const int max_split = T.fuse_wsplit_max;                          // 4; 1 = never cut in width
        for (int ws = 2; shape_ok && !c->fuse12 && band_r2 > 0 && ws <= max_split; ++ws) {
            const int w2p = (l1.OW + ws - 1) / ws, w1m = 2 * w2p + 3;          // every part w2p conv2 columns wide (the last one overlaps its neighbour)
            if (w2p < 15 || w2p > l1.OW || w1m * 19 >= 65536) continue;
            const int cpr = ((2 * w1m + 3) * 3 + 15) / 16;
            for (int r2 = std::min(band_r2, l1.OH); r2 >= std::min(4, l1.OH); --r2) {
                const int rows_in = 2 * (2 * r2 + 3) + 3;
                if (rows_in * cpr > kBandPf * 512) continue;
                int off = 12 * 32 * 16;
                q.off_w2 = off; off += 80 * 32 * 16;
                q.off_b = off; off += 16 * 16;
                q.off_tile = off; q.tile_bytes = (((2 * r2 + 3) * 2 * (w2p + 2) * 48 + 128) + 15) & ~15; off += q.tile_bytes;
                q.off_band = off; q.band_bytes = rows_in * cpr * 32 + 64; off += q.band_bytes;
                if (off > 160 * 1024) continue;
                c->fuse12 = true; c->fuse_r2 = r2; c->fuse_lds = off; q.R2 = r2; q.bands = (l1.OH + r2 - 1) / r2;
                q.wsplit = ws; q.w2p = w2p; q.cpr = cpr;
                q.magic_full = (unsigned)((0x100000000ull + (unsigned)w1m - 1u) / (unsigned)w1m);
                q.magic_cpr = (unsigned)((0x100000000ull + (unsigned)cpr - 1u) / (unsigned)cpr);
                break;
            }
        }
    }
    if (c->fuse12) {                            // the band kernel reads conv1 columns by parity: conv2's granules in that order
        const ConvLayer& l1 = c->L[1];
        std::vector<u4v> orig((size_t)l1.G_pad * l1.COUT_PAD), perm(orig.size());
        HIPCHK(hipMemcpy(orig.data(), l1.w, orig.size() * sizeof(u4v), hipMemcpyDeviceToHost));
        for (int kh = 0; kh < 5; ++kh)
            for (int sl = 0; sl < 16; ++sl) {
                const int src = sl < 9 ? (2 * (sl / 3)) * 3 + sl % 3 : (sl < 15 ? (2 * ((sl - 9) / 3) + 1) * 3 + (sl - 9) % 3 : 15);   // granule kw * 3 + c8 of the kernel row
                std::memcpy(&perm[(size_t)(kh * 16 + sl) * l1.COUT_PAD], &orig[(size_t)(kh * 16 + src) * l1.COUT_PAD], (size_t)l1.COUT_PAD * sizeof(u4v));
            }
        int rcw = upload(&c->w2_parity, perm);
        if (rcw) return rcw;
        c->fuse.w2 = c->w2_parity;
    }
    {   // conv4..conv7 (or conv5..conv7) as one launch when F frames of all their activations fit LDS; TRS_PILOT_CHAIN = 0: off, 3 / 4: layers
        const int want = T.chain_layers;
        c->chain_first = -1;
        for (int nl = std::min(want, 4); nl >= 3 && c->chain_first < 0; --nl) {
            const int first = 7 - nl;
            bool ok = true;
            for (int i = first; i < 7; ++i) {
                const ConvLayer& l = c->L[i];
                ok = ok && l.frame && l.COUT == l.COUT_PAD && l.CIN == (i == 6 ? 128 : 64) && (l.COUT == 64 || l.COUT == 128);
            }
            if (!ok) continue;
            auto out_bytes = [&](int i) { return (size_t)c->L[i].OH * c->L[i].OW * c->L[i].COUT * 2; };
            auto in_bytes_of = [&](int i) { return (size_t)c->L[i].IH * c->L[i].IW * c->L[i].CIN * 2; };
            for (int f = 4; f >= 2 && c->chain_first < 0; f -= 2) {
                if (f > 2 && (c->n_cap + f - 1) / f < c->cu_count) continue;          // keep a workgroup per CU
                const bool split = nl == 4;
                size_t a, b;
                if (split) { a = std::max(f * out_bytes(3), f * out_bytes(5)); b = std::max((size_t)(f / 2) * in_bytes_of(3), f * out_bytes(4)); }
                else { a = std::max(f * in_bytes_of(4), f * out_bytes(5)); b = f * out_bytes(4); }
                a = (a + 15) & ~(size_t)15; b = (b + 15) & ~(size_t)15;
                const size_t total = a + b + 4 * 128 * 4;
                if (total > 158 * 1024) continue;
                ChainParams& q = c->chain;
                q = ChainParams{};
                constexpr int nw = 8;                                               // waves per workgroup, one workgroup per CU
                q.F = f; q.nl = nl; q.split_first = split ? 1 : 0; q.offA = 0; q.offB = (int)a; q.off_bias = (int)(a + b);
                for (int j = 0; j < nl; ++j) {
                    const ConvLayer& l = c->L[first + j];
                    // item shape per layer: items = ceil(pixels / (32 nt)) x (COUT / (32 nb)), dealt round-robin to 8 waves; waves w and w + 4 share
                    // a SIMD (one workgroup per CU): the shape that leaves the busiest SIMD the fewest MFMAs per k-step, and among equals the one
                    // that gives that SIMD two waves (a lone wave per SIMD exposes every LDS and L2 round trip: stamps, profiles/r03_pilot_chain.txt)
                    const int px = (split && j == 0 ? f / 2 : f) * l.OH * l.OW;
                    auto busiest = [&](int nt, int nb, int& waves_on_it) {
                        const int items = ((px + 32 * nt - 1) / (32 * nt)) * (l.COUT / (32 * nb));
                        int worst = 0; waves_on_it = 0;
                        for (int sd = 0; sd < 4; ++sd) {
                            int n_items = 0, n_waves = 0;
                            for (int w = sd; w < nw; w += 4) { const int mine = items > w ? (items - w + nw - 1) / nw : 0; n_items += mine; n_waves += mine > 0; }
                            if (n_items * nt * nb > worst) { worst = n_items * nt * nb; waves_on_it = n_waves; }
                        }
                        return worst;
                    };
                    // the better tile height (2 or 3 tiles of 32 pixels) for items of 64 output channels (items of 32 channels — twice the ring depth,
                    // two waves on every SIMD — were measured in round 3: the kernel 105 k against 107 k clocks, the closed loop equal; removed in round 4, and measured once
                    // more for conv7 alone, whose 64-channel items are 6 for 8 waves or 4 lone waves: chain 46.3 -> 47.5 us, for every layer 49.5; conv7 on ten single-tile
                    // items of 64 channels (three per busy SIMD instead of four lone waves): 43.4 -> 46.5 us — twice the weight stream per MFMA)
                    int nt = 2, best = 1 << 30, best_waves = 0;
                    for (int cnt = 3; cnt >= 2; --cnt) {
                        int wv = 0;
                        const int m = busiest(cnt, 2, wv);
                        if (m < best || (m == best && wv > best_waves)) { best = m; best_waves = wv; nt = cnt; }
                    }
                    auto magic = [](int d) { return (unsigned)((0x100000000ull + (unsigned)d - 1u) / (unsigned)d); };
                    q.L[j] = ChainLayer{l.w, l.bias, l.IH, l.IW, l.OH, l.OW, l.COUT, l.CIN / 8, l.CIN / 8 == 8 ? 3 : 4, nt, 2, magic(l.OH * l.OW), magic(l.OW)};
                }
                c->chain_first = first; c->chain_lds = (int)total;
            }
        }
    }
    HIPCHK(hipMalloc((void**)&c->raw, (size_t)c->n_cap * 2 * sizeof(float)));
    *slot = guard.release();
    return TRS_OK;
}

TRS_EXPORT int trs_pilot_forward(trs_env* e, const uint8_t* d_frames, int n_images, float* d_out)
{
    TrsEnvView v;
    if (!trs_internal_view(e, &v)) return trs_internal_fail(TRS_ERR_ARG, "null handle");
    PilotCtx* c = static_cast<PilotCtx*>(*trs_internal_pilot_slot(e));
    if (!c) return trs_internal_fail(TRS_ERR_STATE, "no pilot loaded");
    if (n_images < 1 || n_images > c->n_cap) return trs_internal_fail(TRS_ERR_ARG, "n_images must be in [1, n_envs]");
    HIPCHK(hipSetDevice(v.device));
    if (!d_frames) {
        if (!v.latest_frame || n_images != v.n) return trs_internal_fail(TRS_ERR_ARG, "latest-frame source needs a rendered step and n_images == n_envs");
        d_frames = v.latest_frame;
    }
    if (c->arch != TRS_PILOT_SPD_CTL && n_images != v.n)
        return trs_internal_fail(TRS_ERR_ARG, "this model type also reads speed (and segment): use trs_pilot_forward_ex, or n_images == n_envs for the env's own");
    return forward_and_tail(c, v, d_frames, n_images, d_out ? d_out : c->raw, nullptr, false);
}

TRS_EXPORT int trs_pilot_forward_ex(trs_env* e, const uint8_t* d_frames, const float* d_speed, const float* d_segment, int n_images, float* d_out)
{
    TrsEnvView v;
    if (!trs_internal_view(e, &v)) return trs_internal_fail(TRS_ERR_ARG, "null handle");
    PilotCtx* c = static_cast<PilotCtx*>(*trs_internal_pilot_slot(e));
    if (!c) return trs_internal_fail(TRS_ERR_STATE, "no pilot loaded");
    if (n_images < 1 || n_images > c->n_cap) return trs_internal_fail(TRS_ERR_ARG, "n_images must be in [1, n_envs]");
    if ((!d_speed || (c->arch == TRS_PILOT_FULL_HOUSE && !d_segment)) && n_images != v.n) return trs_internal_fail(TRS_ERR_ARG, "the env's own speed / segment need n_images == n_envs");
    HIPCHK(hipSetDevice(v.device));
    if (!d_frames) {
        if (!v.latest_frame || n_images != v.n) return trs_internal_fail(TRS_ERR_ARG, "latest-frame source needs a rendered step and n_images == n_envs");
        d_frames = v.latest_frame;
    }
    const ActIo io{d_speed, d_segment, nullptr, nullptr, nullptr, nullptr};
    return forward_and_tail(c, v, d_frames, n_images, d_out ? d_out : c->raw, nullptr, false, &io);
}

TRS_EXPORT int trs_pilot_forward_host_ex(trs_env* e, const uint8_t* h_frames, const float* h_speed, const float* h_segment, int n_images, float* h_out)
{
    TrsEnvView v;
    if (!trs_internal_view(e, &v)) return trs_internal_fail(TRS_ERR_ARG, "null handle");
    PilotCtx* c = static_cast<PilotCtx*>(*trs_internal_pilot_slot(e));
    if (!c) return trs_internal_fail(TRS_ERR_STATE, "no pilot loaded");
    if (!h_frames || !h_out || n_images < 1 || n_images > c->n_cap) return trs_internal_fail(TRS_ERR_ARG, "bad argument (n_images must be in [1, n_envs])");
    if (c->arch != TRS_PILOT_SPD_CTL && (!h_speed || (c->arch == TRS_PILOT_FULL_HOUSE && !h_segment))) return trs_internal_fail(TRS_ERR_ARG, "this model type needs speed (and segment) arrays");
    HIPCHK(hipSetDevice(v.device));
    const size_t bytes = (size_t)n_images * c->H * c->W * 3, extra = (size_t)n_images * 8;
    if (bytes + extra > c->tmp_cap) {
        HIPCHK(hipStreamSynchronize(v.stream));
        (void)hipFree(c->tmp_frames); c->tmp_frames = nullptr; c->tmp_cap = 0;
        HIPCHK(hipMalloc((void**)&c->tmp_frames, bytes + extra + 128));
        c->tmp_cap = bytes + extra;
    }
    float* d_spd = reinterpret_cast<float*>(c->tmp_frames + ((bytes + 63) & ~(size_t)63));
    float* d_seg = d_spd + n_images;
    HIPCHK(hipMemcpyAsync(c->tmp_frames, h_frames, bytes, hipMemcpyHostToDevice, v.stream));
    if (h_speed) HIPCHK(hipMemcpyAsync(d_spd, h_speed, (size_t)n_images * 4, hipMemcpyHostToDevice, v.stream));
    if (h_segment) HIPCHK(hipMemcpyAsync(d_seg, h_segment, (size_t)n_images * 4, hipMemcpyHostToDevice, v.stream));
    int rc = c->arch == TRS_PILOT_SPD_CTL ? trs_pilot_forward(e, c->tmp_frames, n_images, c->raw)
                                          : trs_pilot_forward_ex(e, c->tmp_frames, h_speed ? d_spd : nullptr, h_segment ? d_seg : nullptr, n_images, c->raw);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(h_out, c->raw, (size_t)n_images * 2 * sizeof(float), hipMemcpyDeviceToHost, v.stream));
    trs_internal_count(e, (uint64_t)n_images * 2 * sizeof(float), (uint64_t)bytes);
    HIPCHK(hipStreamSynchronize(v.stream));
    return TRS_OK;
}

TRS_EXPORT int trs_pilot_forward_host(trs_env* e, const uint8_t* h_frames, int n_images, float* h_out)
{
    TrsEnvView v;
    if (!trs_internal_view(e, &v)) return trs_internal_fail(TRS_ERR_ARG, "null handle");
    PilotCtx* c = static_cast<PilotCtx*>(*trs_internal_pilot_slot(e));
    if (!c) return trs_internal_fail(TRS_ERR_STATE, "no pilot loaded");
    if (!h_frames || !h_out || n_images < 1 || n_images > c->n_cap) return trs_internal_fail(TRS_ERR_ARG, "bad argument (n_images must be in [1, n_envs])");
    HIPCHK(hipSetDevice(v.device));
    const size_t bytes = (size_t)n_images * c->H * c->W * 3;
    if (bytes > c->tmp_cap) {
        HIPCHK(hipStreamSynchronize(v.stream));
        (void)hipFree(c->tmp_frames); c->tmp_frames = nullptr; c->tmp_cap = 0;
        HIPCHK(hipMalloc((void**)&c->tmp_frames, bytes + 64));
        c->tmp_cap = bytes;
    }
    HIPCHK(hipMemcpyAsync(c->tmp_frames, h_frames, bytes, hipMemcpyHostToDevice, v.stream));
    int rc = trs_pilot_forward(e, c->tmp_frames, n_images, c->raw);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(h_out, c->raw, (size_t)n_images * 2 * sizeof(float), hipMemcpyDeviceToHost, v.stream));
    trs_internal_count(e, (uint64_t)n_images * 2 * sizeof(float), (uint64_t)n_images * c->H * c->W * 3);
    HIPCHK(hipStreamSynchronize(v.stream));
    return TRS_OK;
}

TRS_EXPORT int trs_pilot_debug_layer(trs_env* e, int layer, float* h_dst, size_t n_floats)
{
    TrsEnvView v;
    if (!trs_internal_view(e, &v)) return trs_internal_fail(TRS_ERR_ARG, "null handle");
    PilotCtx* c = static_cast<PilotCtx*>(*trs_internal_pilot_slot(e));
    if (!c || !c->last_n) return trs_internal_fail(TRS_ERR_STATE, "no forward pass yet");
    if (layer < 0 || layer > 7 || !h_dst) return trs_internal_fail(TRS_ERR_ARG, "bad layer");
    const size_t total = (size_t)c->last_n * c->act_elems[layer];
    if (n_floats != total) return trs_internal_fail(TRS_ERR_ARG, "size mismatch");
    HIPCHK(hipSetDevice(v.device));
    if (layer == 0 && !c->act0_valid) {                                     // the fused head never wrote conv1's activation: run the unfused conv1 now
        int rc = launch_conv(c->L[0], c->last_frames, (size_t)c->last_n * c->H * c->W * 3, c->act[0], c->last_n, v.stream, c->cu_count);
        if (rc) return rc;
        c->act0_valid = true;
    }
    if (c->chain_first >= 0 && layer >= c->chain_first && layer < 6 && !c->chain_mid_valid) {   // the chain kept these activations in LDS: run the single layers now
        for (int j = c->chain_first; j < 6; ++j) {
            const void* src = j == 0 ? (const void*)c->last_frames : c->act[j - 1];
            int rc = launch_conv(c->L[j], src, (size_t)c->last_n * c->act_elems[j - 1] * 2, c->act[j], c->last_n, v.stream, c->cu_count);
            if (rc) return rc;
        }
        c->chain_mid_valid = true;
    }
    HIPCHK(hipStreamSynchronize(v.stream));
    if (c->L[layer].out_f32) {                                              // dense1: add the K-slice slabs in slice order, then the ReLU (both live in the tail kernel)
        std::vector<float> part(total);
        for (size_t i = 0; i < total; ++i) h_dst[i] = 0.0f;
        for (int sl = 0; sl < c->last_slices; ++sl) {
            HIPCHK(hipMemcpy(part.data(), static_cast<const float*>(c->slab) + (size_t)sl * total, total * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < total; ++i) h_dst[i] += part[i];
        }
        for (size_t i = 0; i < total; ++i) h_dst[i] = h_dst[i] > 0.0f ? h_dst[i] : 0.0f;
        return TRS_OK;
    }
    std::vector<unsigned short> tmp(total);
    HIPCHK(hipMemcpy(tmp.data(), c->act[layer], total * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < total; ++i) h_dst[i] = host_h2f(tmp[i]);
    return TRS_OK;
}

// materialise what the fused kernels of the last forward pass kept in LDS (conv1 behind the fused head, the chain's interior layers)
static int materialise_layers(PilotCtx* c, const TrsEnvView& v, int upto)
{
    if (!c->act0_valid) {
        int rc = launch_conv(c->L[0], c->last_frames, (size_t)c->last_n * c->H * c->W * 3, c->act[0], c->last_n, v.stream, c->cu_count);
        if (rc) return rc;
        c->act0_valid = true;
    }
    if (c->chain_first >= 0 && upto >= c->chain_first && !c->chain_mid_valid) {
        for (int j = c->chain_first; j < 6; ++j) {
            int rc = launch_conv(c->L[j], c->act[j - 1], (size_t)c->last_n * c->act_elems[j - 1] * 2, c->act[j], c->last_n, v.stream, c->cu_count);
            if (rc) return rc;
        }
        c->chain_mid_valid = true;
    }
    return TRS_OK;
}

TRS_EXPORT int trs_pilot_range_check(trs_env* e, uint64_t h_out[8])
{
    TrsEnvView v;
    if (!trs_internal_view(e, &v)) return trs_internal_fail(TRS_ERR_ARG, "null handle");
    PilotCtx* c = static_cast<PilotCtx*>(*trs_internal_pilot_slot(e));
    if (!c || !c->last_n) return trs_internal_fail(TRS_ERR_STATE, "no forward pass yet");
    if (!h_out) return trs_internal_fail(TRS_ERR_ARG, "null output");
    HIPCHK(hipSetDevice(v.device));
    { int rc = materialise_layers(c, v, 6); if (rc) return rc; }
    unsigned long long* const scratch = v.stats + 24;                       // stats[24..30]: this call's per-layer counts; stats[3]: running total
    HIPCHK(hipMemsetAsync(scratch, 0, 8 * sizeof(unsigned long long), v.stream));
    for (int i = 0; i < 7; ++i) {
        const size_t n = (size_t)c->last_n * c->act_elems[i];
        const int grid = (int)std::min<size_t>(4096, (n + 255) / 256);
        hipLaunchKernelGGL(trs_pilot_count_sat_kernel, dim3(grid), dim3(256), 0, v.stream, static_cast<const unsigned short*>(c->act[i]), n, scratch + i, v.stats + 3);
    }
    HIPCHK(hipGetLastError());
    unsigned long long host[8] = {};
    HIPCHK(hipMemcpyAsync(host, scratch, sizeof host, hipMemcpyDeviceToHost, v.stream));
    HIPCHK(hipStreamSynchronize(v.stream));
    host[7] = 0;
    for (int i = 0; i < 7; ++i) { h_out[i] = host[i]; host[7] += host[i]; }
    h_out[7] = host[7];
    return TRS_OK;
}

TRS_EXPORT int trs_pilot_act(trs_env* e, const trs_pilot_config* cfg, const uint8_t* d_frames, const float* d_speed, const float* d_segment,
                             const uint8_t* d_mode, float* d_steer, float* d_thr, float* d_brk, int n)
{
    TrsEnvView v;
    if (!trs_internal_view(e, &v)) return trs_internal_fail(TRS_ERR_ARG, "null handle");
    PilotCtx* c = static_cast<PilotCtx*>(*trs_internal_pilot_slot(e));
    if (!c) return trs_internal_fail(TRS_ERR_STATE, "no pilot loaded");
    if (!cfg || cfg->struct_size != sizeof(trs_pilot_config)) return trs_internal_fail(TRS_ERR_ARG, "trs_pilot_config.struct_size mismatch");
    { int rm = check_model_type(c, cfg); if (rm) return rm; }
    if (!d_steer || !d_thr || !d_brk || n < 0 || n > c->n_cap) return trs_internal_fail(TRS_ERR_ARG, "null output or n out of range (n <= n_envs)");
    if ((!d_speed || (c->arch == TRS_PILOT_FULL_HOUSE && !d_segment)) && n != v.n) return trs_internal_fail(TRS_ERR_ARG, "the env's own speed / segment need n == n_envs");
    HIPCHK(hipSetDevice(v.device));
    if (n == 0) return TRS_OK;
    if (!d_frames) {
        if (n != v.n) return trs_internal_fail(TRS_ERR_ARG, "latest-frame source needs n == n_envs");
        d_frames = v.latest_frame;
    }
    if (!d_frames) {                                // args[0] is None -> (0.0, 0.0, 0.0) (keras_pilot.py:46-47)
        hipLaunchKernelGGL(trs_zero_controls_kernel, dim3((n + 255) / 256), dim3(256), 0, v.stream, d_steer, d_thr, d_brk, n);
        HIPCHK(hipGetLastError());
        return TRS_OK;
    }
    const ActIo io{d_speed, d_segment, d_mode, d_steer, d_thr, d_brk};
    return forward_and_tail(c, v, d_frames, n, c->raw, cfg, true, &io);
}

TRS_EXPORT int trs_step_pilot(trs_env* e, const trs_pilot_config* cfg, int n_steps)
{
    TrsEnvView v;
    if (!trs_internal_view(e, &v)) return trs_internal_fail(TRS_ERR_ARG, "null handle");
    PilotCtx* c = static_cast<PilotCtx*>(*trs_internal_pilot_slot(e));
    if (!c) return trs_internal_fail(TRS_ERR_STATE, "no pilot loaded");
    if (!cfg || cfg->struct_size != sizeof(trs_pilot_config)) return trs_internal_fail(TRS_ERR_ARG, "trs_pilot_config.struct_size mismatch");
    if (n_steps < 1) return trs_internal_fail(TRS_ERR_ARG, "n_steps < 1");
    { int rm = check_model_type(c, cfg); if (rm) return rm; }
    if (!v.render) return trs_internal_fail(TRS_ERR_STATE, "the pilot needs a camera (render = 1)");
    HIPCHK(hipSetDevice(v.device));
    for (int k = 0; k < n_steps; ++k) {
        trs_internal_view(e, &v);
        if (v.latest_frame) {                       // KerasPilot.step on the frame of the previous tick
            int rc = forward_and_tail(c, v, v.latest_frame, v.n, c->raw, cfg, true);
            if (rc) return rc;
        } else {                                    // args[0] is None -> (0.0, 0.0, 0.0) (keras_pilot.py:46-47)
            hipLaunchKernelGGL(trs_zero_controls_kernel, dim3((v.n + 255) / 256), dim3(256), 0, v.stream, v.ctl_steer, v.ctl_thr, v.ctl_brk, v.n);
            HIPCHK(hipGetLastError());
        }
        int rc = trs_internal_step_launch(e, v.ctl_steer, v.ctl_thr, v.ctl_brk);
        if (rc) return rc;
    }
    return TRS_OK;
}
