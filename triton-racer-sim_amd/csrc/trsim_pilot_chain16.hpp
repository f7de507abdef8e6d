// trsim_pilot_chain16.hpp — conv4 .. conv7 in one launch on v_mfma_f32_16x16x32_f16 (round 5), included by trsim_pilot.hip.
//
// What it replaces: the 3x3 'valid' + ReLU layers conv4..conv7 of Keras_2D_CNN.get_model (reference components/keras_train.py:143-156).
// Same scheme as trs_conv_chain_kernel (a frame's activations never leave LDS, weights straight from L2 through a register ring, the K loop one
// basic block), rebuilt around the 16x16x32 MFMA for three measured reasons:
//   * clock: an MFMA-dense loop of the 32x32x16 shape holds ~1.62 GHz on this chip, of the 16x16x32 shape ~2.05 GHz — 1.70 against 1.96-2.00 PFLOP/s
//     with the operands in registers, on random data (profiles/r05_mfma_issue.txt; /opt/skills/guides/MI355X_MICROARCH.md, DVFS item 7);
//   * tile quantisation: the old items are 64 or 96 pixels x 64 channels, and these layers are small (conv7: 36 pixels per frame): 21 % of the MFMA
//     cycles of the old chain were padding (profiles/r04_pilot_pmc.txt).  Here an item is PB blocks of 16 pixels x CB blocks of 16 channels, chosen per
//     layer so that the 8 waves get equal shares;
//   * LDS bank conflicts: 46 % of the old chain's LDS cycles.  A tile of consecutive output pixels crosses image rows, and behind a
//     row wrap the input pixels of its lanes no longer sit on distinct banks.  Here (a) the LDS image of an activation is TWO planes — the even and the
//     odd 16-byte channel granules of every pixel — because the 16x16x32 operand layout makes the two 16-lane halves of a ds_read_b128 group read granules
//     g and g + 1 of different pixels: with both in one plane no swizzle is conflict-free for every tap shift (proof by exhaustion in
//     tests/test_pilot_layout.py); (b) WHICH pixel a column of a block holds is a host-built table: column i of every block holds a pixel whose
//     linear index in the LDS image is = i (mod 16), so every tap (a constant shift of all indices) reads 16 distinct bank quads per group.
//     conflict-free by construction for every tap, k-step, alignment and row wrap (same test).
// Operands: A = weights (rows = 16 output channels, k = 32 input channels of one tap = 4 granules, lane l: channel l % 16, granule l / 16), straight
// from the layer's [9 taps x cg granules][COUT] image by buffer load; B = pixels (columns = 16 pixels, lane l: pixel l % 16, granule l / 16) from LDS;
// D: lane l holds channels 4 (l / 16) .. + 3 of pixel l % 16.  fp16 operands, fp32 accumulate from the bias, ReLU + fp16 (saturating) in the epilogue.
#pragma once

typedef __attribute__((ext_vector_type(4))) float f32x4;
#ifndef TRS_C16_ABLATE
#define TRS_C16_ABLATE 0   /* timing-only diagnostic builds, never shipped (results wrong): 1 = no weight refills, 2 = pixel fragments of the first k-step only, 3 = no MFMA */
#endif

struct Chain16Layer {
    const u4v* w; const float* bias;
    const u4v* cols;                       // [n_pgroups * pb][16]: x = linear pixel index of the column's window (tap 0) in the layer's INPUT image, y = output pixel slot
                                           // (ul * OH + oy) * OW + ox (0xffffffff: none), z = that pixel's linear index in the NEXT layer's input image (unit 0 = the pass's first)
    int IW, IHW, OHW, COUT, cg, cgs;       // input row pitch in pixels, input / output pixels per unit, channels, input granules per pixel (8 / 16) and log2
    int unit_in;                           // pixels between the units (frames) of the INPUT image: IHW + a few pixels of padding, chosen by the host so that the
                                           // window starts of a unit's output pixels spread evenly over the 16 residue classes (blocks = ceil(pixels / 16), not more)
    int pb, cb, n_pgroups, n_cgroups;      // a wave item = pb blocks of 16 pixels x cb blocks of 16 channels (cb even)
    int plane_in;                          // byte distance between the even-granule and the odd-granule plane of this layer's INPUT image
};
struct Chain16Params {
    const u4v* in;                         // the first layer's input activation, fp16 NHWC
    unsigned short* out;                   // the last layer's output activation
    int N, F, nl, split_first;
    int offA, offB, off_bias;
    Chain16Layer L[4];
};

// granule swizzle inside a plane: a plane holds cg / 2 granules per pixel (64 B at 64 channels: four pixels per 256-B bank row; 128 B at 128: two)
__device__ __forceinline__ int plane_swz(int pix, int cgs) { return cgs == 3 ? ((pix >> 2) & 3) : ((pix >> 1) & 7); }
// 16-byte slot of (pixel, granule q) in a two-plane image, relative to the image base, in bytes
__device__ __forceinline__ unsigned plane_slot(int pix, int q, int cgs, int plane_bytes)
{
    return (unsigned)((q & 1) * plane_bytes) + (unsigned)(((pix << (cgs - 1)) + ((q >> 1) ^ plane_swz(pix, cgs))) << 4);
}

// One layer.  KS = k-steps of 32 input channels per tap (2 at 64 channels, 4 at 128); COUT = the layer's output channels (a template parameter: a k-step's
// weight offset k x 4 x COUT x 16 is then a literal of the instruction stream — a running scalar pinned by an asm constraint, as the 32x32 kernels have it, cannot
// be allocated in the tall instantiations, and offsets computed from a run-time COUT get hoisted out of the K loop and spilled).
template <int KS, int COUT, int R, int PB, int CB>
__device__ __forceinline__ void chain16_layer(const Chain16Layer& L, unsigned lin_b, const float* lbias_f, int nu, bool to_lds, unsigned lout_b, int cgs_out, int plane_out, int out_pix0,
                                              unsigned short* gout, int wave, int nwaves, int lane)   // out_pix0: the pass's first unit in the next layer's image, in pixels
{
    static_assert(CB % 2 == 0, "channel blocks come in pairs (the epilogue's half exchange)");
    const int c16 = lane & 15, kb = lane >> 4;
    const int m_wg = nu * L.OHW;
    constexpr int CGS = KS == 2 ? 3 : 4, ksteps = 9 * KS;
    const float4* lbias = reinterpret_cast<const float4*>(lbias_f);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<u4v*>(L.w), 0, 9 * L.cg * COUT * 16, 0x00020000);
    const unsigned plane_b = lin_b + (unsigned)((kb & 1) * L.plane_in);              // this lane's plane: granule 4 kk + kb is odd iff kb is
    for (int item = wave; item < L.n_pgroups * L.n_cgroups; item += nwaves) {
        const int cgrp = item / L.n_pgroups, pg = item - cgrp * L.n_pgroups;
        const int cbase = cgrp * CB * 16;
        u4v ent[PB];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) ent[pb] = L.cols[(pg * PB + pb) * 16 + c16];
        // weight granule (4 kk + kb, channel cbase + cb * 16 + c16): one lane offset, the block as the immediate, ONE scalar that runs with the loads
        const int wvoff = (kb * COUT + cbase + c16) * 16;
        auto wload = [&](int k, int cb) { return __builtin_bit_cast(u4v, __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff + cb * 256, k * (4 * COUT * 16), 0)); };   // (k: compile-time)
        u4v ring[R][CB];
#pragma unroll
        for (int d = 0; d < R; ++d)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) ring[d][cb] = wload(d, cb);
        f32x4 acc[PB][CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const float4 b = lbias[(cbase + cb * 16 + 4 * kb) >> 2];
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) { acc[pb][cb][0] = b.x; acc[pb][cb][1] = b.y; acc[pb][cb][2] = b.z; acc[pb][cb][3] = b.w; }
        }
        // LDS byte address of this lane's granule of the current tap's pixel, k-step 0: granule-in-plane (2 kk + (kb >> 1)) ^ swizzle = ((kb >> 1) ^ swizzle) ^ 2 kk:
        // the tap's address ^ 32 kk, one v_xor per k-step and block (a K loop is priced by its non-MFMA instructions: profiles/r04_mfma_issue.txt)
        unsigned tapb[PB];
        auto pixels = [&](int k, h16x8 (&x)[PB]) {                           // k is a compile-time constant after unrolling
            const int tap = k / KS, kk = k % KS;
            const int tap_off = (tap / 3) * L.IW + tap % 3;
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                if (kk == 0) {
                    const int pix = (int)ent[pb].x + tap_off;
                    tapb[pb] = plane_b + (unsigned)(((pix << (CGS - 1)) + ((kb >> 1) ^ plane_swz(pix, CGS))) << 4);
                }
                x[pb] = __builtin_bit_cast(h16x8, *(lds_u4vp)(uintptr_t)(tapb[pb] ^ (unsigned)(32 * kk)));
            }
        };
        h16x8 xa[PB], xb[PB];
        pixels(0, xa);
#pragma unroll
        for (int k = 0; k < ksteps; ++k) {
            const int d = k % R;
            h16x8 (&xc)[PB] = (k & 1) ? xb : xa;
            h16x8 (&xn)[PB] = (k & 1) ? xa : xb;
            if (k + 1 < ksteps && (TRS_C16_ABLATE != 2 || k == 0)) pixels(k + 1, xn);
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
#if TRS_C16_ABLATE == 3
                    asm volatile("" :: "v"(ring[d][cb]), "v"(xc[pb]));
#else
                    acc[pb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, ring[d][cb]), (TRS_C16_ABLATE == 2 && k > 1) ? ((k & 1) ? xa[pb] : xb[pb]) : xc[pb], acc[pb][cb], 0, 0, 0);
#endif
                }
            if (k + R < ksteps && TRS_C16_ABLATE != 1) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) ring[d][cb] = wload(k + R, cb);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // Epilogue.  Lane (c16, kb) holds channels cb * 16 + 4 kb .. + 3 of its pixel for every block cb: half a 16-byte granule.  For a PAIR of blocks
        // (cb0, cb1) v_permlane16_swap_b32 exchanges the odd 16-lane rows of the first operand with the even rows of the second: an even-kb lane ends with
        // (its own, its partner's) half of the granule of block cb0, an odd-kb lane with (its partner's, its own) half of the granule of block cb1 — every
        // lane stores ONE whole granule per pair, two full-rate VALU instructions per pair and pixel block.
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            const int m = (int)ent[pb].y;
            const bool valid = (unsigned)m < (unsigned)m_wg;
#pragma unroll
            for (int cp = 0; cp < CB; cp += 2) {
                uint2 w0 = relu_pack4(acc[pb][cp][0], acc[pb][cp][1], acc[pb][cp][2], acc[pb][cp][3]);
                uint2 w1 = relu_pack4(acc[pb][cp + 1][0], acc[pb][cp + 1][1], acc[pb][cp + 1][2], acc[pb][cp + 1][3]);
                { const auto r = __builtin_amdgcn_permlane16_swap(w0.x, w1.x, false, false); w0.x = r[0]; w1.x = r[1]; }
                { const auto r = __builtin_amdgcn_permlane16_swap(w0.y, w1.y, false, false); w0.y = r[0]; w1.y = r[1]; }
                const u4v g = u4v{w0.x, w0.y, w1.x, w1.y};
                const int q = ((cbase + (cp + (kb & 1)) * 16) >> 3) + (kb >> 1);     // the granule (8 channels) this lane holds now
                if (valid) {
                    if (to_lds) {                                               // the next layer's two-plane image
                        const int po = out_pix0 + (int)ent[pb].z;
                        *(__attribute__((address_space(3))) u4v*)(uintptr_t)(lout_b + plane_slot(po, q, cgs_out, plane_out)) = g;
                    } else {
                        *reinterpret_cast<u4v*>(gout + (size_t)m * COUT + 8 * q) = g;
                    }
                }
            }
        }
    }
}

#ifndef TRS_C16_STAMPS
#define TRS_C16_STAMPS 0   /* diagnostic build, never shipped: workgroup 7's waves print the s_memtime ticks of their phases behind the last barrier */
#endif
#if TRS_C16_STAMPS
#define C16_STAMP(i) do { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); st_[i] += t_ - t0_; t0_ = t_; } while (0)
#else
#define C16_STAMP(i) do { } while (0)
#endif
#ifndef TRS_CHAIN16_R
#define TRS_CHAIN16_R 3   /* weight ring depth in k-steps of 32 channels (a k-step is pb x cb x 16 MFMA cycles: 300-550 at the shapes below) */
#endif
// pixel blocks of a layer's items -> the instantiation (channel blocks: always 2 = 32 channels).  Only these heights exist (each is a fully unrolled K loop:
// compile time); the host's plan picks among them (kChain16Pb).
template <int KS, int COUT>
__device__ __forceinline__ void chain16_layer_any(const Chain16Layer& L, unsigned lin_b, const float* lbias_f, int nu, bool to_lds, unsigned lout_b, int cgs_out, int plane_out, int out_pix0,
                                                  unsigned short* gout, int wave, int nwaves, int lane)
{
#define TRS_C16(P) case P: chain16_layer<KS, COUT, TRS_CHAIN16_R, P, 2>(L, lin_b, lbias_f, nu, to_lds, lout_b, cgs_out, plane_out, out_pix0, gout, wave, nwaves, lane); break;
    switch (L.pb) { TRS_C16(2) TRS_C16(4) TRS_C16(5) TRS_C16(6) TRS_C16(7) TRS_C16(8) TRS_C16(9) TRS_C16(10) TRS_C16(12) TRS_C16(13) TRS_C16(15) TRS_C16(17) default: break; }
#undef TRS_C16
}
constexpr int kChain16Pb[] = {2, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17};
#ifndef TRS_C16_WAVES
#define TRS_C16_WAVES 4   /* waves per workgroup: 4 = one per SIMD (tall items), 8 = two per SIMD (items half as tall) */
#endif

// ONE wave per SIMD (4 waves per workgroup, one workgroup per CU): a wave may use all 512 registers of its SIMD lane, so an item can be 9-17 pixel blocks
// by 2 channel blocks — a weight fragment fetched from L2 feeds that many MFMAs.  With 8 waves and items half as tall (the first form of this kernel, and the
// 32x32 chain) every weight byte went through the CU's L1 twice to four times per layer: 16-32 KB per k-step at the L1's 64 B/clk = 60-80 % of the k-step's
// MFMA time, which is what held both chains at ~50 % of the MFMA rate (45 us against 43 us: profiles/r05_pilot_chain16.txt).  A k-step of one wave is
// 300-550 MFMA cycles here, so one LDS round trip of prefetch and a ring of 3 k-steps of weights cover their latencies without a second wave.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK, 1) void trs_conv_chain16_kernel(const Chain16Params p)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int nwaves = BLOCK / 64;
    const int u0 = blockIdx.x * p.F, nu = min(p.F, p.N - u0);
    float* const lb = reinterpret_cast<float*>(psmem + p.off_bias);         // [layer][128]
    for (int li = 0; li < p.nl; ++li)
        for (int i = tid; i < p.L[li].COUT; i += BLOCK) lb[li * 128 + i] = p.L[li].bias[i];
    const unsigned lds0 = (unsigned)(uintptr_t)psmem;
    const unsigned A = lds0 + (unsigned)p.offA, B = lds0 + (unsigned)p.offB;
#if TRS_C16_STAMPS
    long long st_[16] = {}, t0_ = (long long)__builtin_amdgcn_s_memtime();
#endif
    // frames fa .. fa + cnt - 1 of the first layer's input into a two-plane image at lds_byte (LDS-DMA: the LDS side is lane-linear, the plane and the
    // swizzle go on the SOURCE address): slot s of plane b holds granule 2 ((s & (cg/2 - 1)) ^ swizzle) + b of image pixel s >> (cgs - 1); the units of the image
    // are unit_in pixels apart (the padding pixels between them are never read by a valid column)
    auto stage = [&](const Chain16Layer& L, int fa, int cnt, unsigned lds_byte) {
        const int per_unit = L.IHW << (L.cgs - 1);                          // granule slots of one unit in one plane
        for (int ul = 0; ul < cnt; ++ul) {
            const u4v* src = p.in + ((size_t)(fa + ul) * L.IHW << L.cgs);
            const int pix0 = ul * L.unit_in;
            for (int b = 0; b < 2; ++b)
                for (int s0 = wave * 64; s0 < per_unit; s0 += nwaves * 64) {
                    const int sl = s0 + lane;
                    if (sl < per_unit) {
                        const int pl = sl >> (L.cgs - 1), gq = sl & ((L.cg >> 1) - 1);     // pixel of the unit, granule slot of the plane
                        const int g = 2 * (gq ^ plane_swz(pix0 + pl, L.cgs)) + b;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (pl << L.cgs) + g),
                                                         (__attribute__((address_space(3))) void*)(uintptr_t)(lds_byte + (unsigned)(b * L.plane_in) + (unsigned)((pix0 << (L.cgs - 1)) + s0) * 16u), 16, 0, 0);
                    }
                }
        }
    };
    int li = 0;
    unsigned cur = A;
    if (p.split_first) {
        const Chain16Layer& L0 = p.L[0];
        const int per = p.F / 2;
        for (int pass = 0; pass < 2; ++pass) {
            const int cnt = min(per, nu - per * pass);                      // workgroup-uniform
            if (cnt > 0) stage(L0, u0 + per * pass, cnt, B);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            C16_STAMP(0);                                                   // staging issued and landed (this wave's part)
            __syncthreads();
            C16_STAMP(1);                                                   // barrier behind the staging
            if (cnt > 0) chain16_layer_any<2, 64>(L0, B, lb, cnt, true, A, p.L[1].cgs, p.L[1].plane_in, per * pass * p.L[1].unit_in, nullptr, wave, nwaves, lane);
            C16_STAMP(2 + pass);                                            // conv4 pass
            __syncthreads();
            C16_STAMP(4);                                                   // barriers behind the layers
        }
        li = 1;
    } else {
        stage(p.L[0], u0, nu, A);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    for (; li < p.nl - 1; ++li) {
        const Chain16Layer& L = p.L[li];
        const unsigned nxt = cur == A ? B : A;
        if (L.COUT == 64) chain16_layer_any<2, 64>(L, cur, lb + li * 128, nu, true, nxt, p.L[li + 1].cgs, p.L[li + 1].plane_in, 0, nullptr, wave, nwaves, lane);
        else chain16_layer_any<2, 128>(L, cur, lb + li * 128, nu, true, nxt, p.L[li + 1].cgs, p.L[li + 1].plane_in, 0, nullptr, wave, nwaves, lane);
        C16_STAMP(5 + li);
        __syncthreads();
        C16_STAMP(4);
        cur = nxt;
    }
    {
        const Chain16Layer& L = p.L[p.nl - 1];
        chain16_layer_any<4, 128>(L, cur, lb + (p.nl - 1) * 128, nu, false, 0u, 0, 0, 0, p.out + (size_t)u0 * L.OHW * L.COUT, wave, nwaves, lane);
        C16_STAMP(9);
    }
#if TRS_C16_STAMPS
    __syncthreads();
    if (blockIdx.x == 7 && lane == 0)
        printf("chain16 workgroup 7 wave %d [ticks]: staging %lld | barrier %lld | conv4 %lld + %lld | conv5 %lld | conv6 %lld | conv7 %lld | layer barriers %lld   (plans pb: %d %d %d %d)\n",
               wave, st_[0], st_[1], st_[2], st_[3], st_[6], st_[7], st_[9], st_[4], p.L[0].pb, p.L[1].pb, p.L[2].pb, p.L[3].pb);
#endif
}
