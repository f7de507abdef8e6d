// trsim_pilot_layers.hpp — the pilot's SINGLE-LAYER convolution kernels (included by trsim_pilot.hip inside its anonymous namespace).
//
// The closed loop's default shapes run on the fused kernels of trsim_pilot.hip (trs_conv12_band_kernel, trs_conv_frame5_kernel,
// trs_conv_chain_kernel / trs_conv_frame_kernel, trs_pilot_dense_kernel).  The kernels here serve what those do not:
//   trs_conv_u8_kernel    conv1 as its own layer (trs_pilot_tuning.no_fuse, a frame shape whose band does not fit LDS, and the activation the
//                         fused head never writes: trs_pilot_debug_layer(0), trs_pilot_range_check)
//   trs_conv_span_kernel  the stride-2 5x5 layers from per-row input spans staged in LDS: conv2 unfused, conv3 where trs_conv_frame5_kernel does not apply
//                         (trs_pilot_tuning.frame5 = 0; until round 4 conv3's default at 240x320 — row bands on the frame5 kernel are faster now)
//   trs_conv_lt_kernel    every other (layer, shape): quad-coalesced pixel loads + LDS transpose — the one generic fallback
// Round 4 removed the chunked kernel (trs_conv_mfma_kernel: rounds 1-2's dense1 and last-resort fallback) and the direct form of the fused head
// (trs_conv12_kernel), which no default shape reached.
#pragma once

// pixel index -> byte offset of its input window.  The tile's first pixel is wave-uniform, so its frame / remainder split
// runs on the scalar unit; a lane adds its own offset (at most a few wraps) and divides the in-frame remainder by OW with a
// float reciprocal + correction (remainders are far below 2^22).
__device__ __forceinline__ int window_base(const ConvParams& p, int n0, int rem0, int add, float inv_ow)
{
    const int ohw = p.OH * p.OW;
    int n = n0, rem = rem0 + add;
    while (rem >= ohw) { rem -= ohw; ++n; }
    int oy = (int)(((float)rem + 0.5f) * inv_ow);
    int ox = rem - oy * p.OW;
    if (ox < 0) { --oy; ox += p.OW; } else if (ox >= p.OW) { ++oy; ox -= p.OW; }
    return ((n * p.IH + oy * p.S) * p.IW + ox * p.S) * p.in_px_bytes;
}

// Epilogue of a 32-pixel tile: bias + ReLU + fp16, transposed through the wave's 2 KB LDS stage so that the global stores
// are 16 bytes per lane and contiguous across lanes (the direct 8-byte stores of the C/D layout cost the addresser one
// lookup per lane: 31 us of conv2's 131).  Stage layout: [pixel][16-B chunk ^ f(pixel)], f spreads the 16 lanes of a
// ds_write_b64 / ds_read_b128 group over the bank row.  NB = 2 (128 B per pixel) goes in two passes of 16 pixels.
template <int NB>
__device__ __forceinline__ void store_tile_at(u4v* stage, const f32x16 (&acc)[NB], const float4* lbias, const ConvParams& p, int m_base, int m_limit, int cbase, int lane)
{   // the tile's 32 pixels are output pixels m_base .. m_base + 31 (consecutive in memory); those >= m_limit are not stored
    constexpr int CR = NB * 4;                                              // 16-B chunks per pixel row of this slice
    constexpr int PP = 128 / CR;                                            // pixels per pass (2 KB stage)
    const int r = lane & 31, h = lane >> 5;
    const int crv = min(CR, (p.COUT - cbase) >> 3);                         // valid chunks (conv1: 3 of 4)
    const unsigned magic = (65536u + (unsigned)crv - 1u) / (unsigned)crv;   // g / crv for g < 256
    uint2* st2 = reinterpret_cast<uint2*>(stage);
#pragma unroll
    for (int pass = 0; pass < 32 / PP; ++pass) {
        const int pr = r - pass * PP;                                       // pixel of this lane within the pass
        if (pr >= 0 && pr < PP) {
            const int f = NB == 1 ? (pr >> 1) & 3 : pr & 7;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b = lbias[nb * 8 + 2 * q + h];
                    const float os = p.oscale;
                    float v0 = __builtin_fmaf(acc[nb][4 * q], os, b.x), v1 = __builtin_fmaf(acc[nb][4 * q + 1], os, b.y), v2 = __builtin_fmaf(acc[nb][4 * q + 2], os, b.z), v3 = __builtin_fmaf(acc[nb][4 * q + 3], os, b.w);
                    st2[(pr * CR + ((4 * nb + q) ^ f)) * 2 + h] = p.relu ? relu_pack4(v0, v1, v2, v3) : make_uint2(pack_h16x2(v0, v1), pack_h16x2(v2, v3));
                }
            }
        }
        const int total = PP * crv;                                         // 16-B chunks to write out in this pass
        const size_t pass_byte0 = (size_t)(m_base + pass * PP) * p.COUT * 2;
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(static_cast<unsigned char*>(p.out) + pass_byte0, 0, PP * p.COUT * 2, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int g = i * 64 + lane;
            if (g < total) {
                const int pix = (int)(((unsigned)g * magic) >> 16), c = g - pix * crv;
                const int m = m_base + pass * PP + pix;
                const int f = NB == 1 ? (pix >> 1) & 3 : pix & 7;
                if (m < m_limit) {
                    const u4v v = stage[pix * CR + (c ^ f)];
                    // a buffer store so that the cache policy can be chosen per layer (immediate aux bits): activations far
                    // larger than L2 leave non-temporally and do not displace what the next layer is about to read
                    const int off = (pix * p.COUT + cbase + 8 * c) * 2;
                    if (p.nt_out == 1) __builtin_amdgcn_raw_buffer_store_b128(v, rout, off, 0, 2);         // nt
                    else if (p.nt_out == 2) __builtin_amdgcn_raw_buffer_store_b128(v, rout, off, 0, 17);   // sc0 sc1 (write-through)
                    else __builtin_amdgcn_raw_buffer_store_b128(v, rout, off, 0, 0);
                }
            }
        }
    }
}

template <int NB>
__device__ __forceinline__ void store_tile(u4v* stage, const f32x16 (&acc)[NB], const float4* lbias, const ConvParams& p, int tile, int cbase, int lane)
{
    store_tile_at<NB>(stage, acc, lbias, p, tile * 32, p.M, cbase, lane);
}

// The epilogue without LDS (round 4; the fused head's conv2 and the frame kernels use the same): the accumulators STARTED at the bias (acc_bias),
// so what is left is ReLU + fp16, the exchange of channel quads between the two lanes of a pixel (v_permlane32_swap_b32) and two 16-byte
// stores per lane and 32-channel block (couts cbase + nb*32 + 16 h .. + 15).  The staged epilogue above costs a tile six dependent LDS round
// trips (four broadcast bias reads + the transpose): 1.4 k clocks in the head (profiles/r04_pilot_head_stamps.txt).  For whole 32-channel blocks
// (COUT % 32 == 0) and oscale == 1 only: the span kernel's layers.
template <int NB>
__device__ __forceinline__ void acc_bias(f32x16 (&acc)[NB], const float4* lbias, int h)
{
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b = lbias[nb * 8 + 2 * q + h];
            acc[nb][4 * q] = b.x; acc[nb][4 * q + 1] = b.y; acc[nb][4 * q + 2] = b.z; acc[nb][4 * q + 3] = b.w;
        }
}
template <int NB>
__device__ __forceinline__ void store_tile_direct(const f32x16 (&acc)[NB], const ConvParams& p, const __amdgpu_buffer_rsrc_t rout, int tile, int cbase, int lane)
{
    const int r = lane & 31, h = lane >> 5, m = tile * 32 + r;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        uint2 w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            w[q] = p.relu ? relu_pack4(acc[nb][4 * q], acc[nb][4 * q + 1], acc[nb][4 * q + 2], acc[nb][4 * q + 3])
                          : make_uint2(pack_h16x2(acc[nb][4 * q], acc[nb][4 * q + 1]), pack_h16x2(acc[nb][4 * q + 2], acc[nb][4 * q + 3]));
        u4v g0, g1;
        quad_groups(w, g0, g1);
        if (m < p.M) {
            const int off = (m * p.COUT + cbase + nb * 32 + 16 * h) * 2;
            if (p.nt_out == 1) { __builtin_amdgcn_raw_buffer_store_b128(g0, rout, off, 0, 2); __builtin_amdgcn_raw_buffer_store_b128(g1, rout, off + 16, 0, 2); }          // nt
            else if (p.nt_out == 2) { __builtin_amdgcn_raw_buffer_store_b128(g0, rout, off, 0, 17); __builtin_amdgcn_raw_buffer_store_b128(g1, rout, off + 16, 0, 17); }   // sc0 sc1
            else { __builtin_amdgcn_raw_buffer_store_b128(g0, rout, off, 0, 0); __builtin_amdgcn_raw_buffer_store_b128(g1, rout, off + 16, 0, 0); }
        }
    }
}

// The same epilogue for a tile whose 32 pixels are a run of a ROW-SEGMENT grid (the fused head's band cut in width: rows of w2
// pixels inside an output activation of OW pixels per row): pixel pg of the band part = (row pg / w2, column pg % w2), output
// pixel m_row0 + row * OW + column.  32 output channels (NB = 1).
__device__ __forceinline__ void store_tile_rows(u4v* stage, const f32x16 (&acc)[1], const float4* lbias, const ConvParams& p, int pg0, int npx, int w2, float inv_w2,
                                                int m_row0, int OW, int lane)
{
    constexpr int CR = 4;
    const int r = lane & 31, h = lane >> 5;
    uint2* st2 = reinterpret_cast<uint2*>(stage);
    {
        const int f = (r >> 1) & 3;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b = lbias[2 * q + h];
            const float os = p.oscale;
            float v0 = __builtin_fmaf(acc[0][4 * q], os, b.x), v1 = __builtin_fmaf(acc[0][4 * q + 1], os, b.y), v2 = __builtin_fmaf(acc[0][4 * q + 2], os, b.z), v3 = __builtin_fmaf(acc[0][4 * q + 3], os, b.w);
            st2[(r * CR + (q ^ f)) * 2 + h] = p.relu ? relu_pack4(v0, v1, v2, v3) : make_uint2(pack_h16x2(v0, v1), pack_h16x2(v2, v3));
        }
    }
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(static_cast<unsigned char*>(p.out), 0, p.M * p.COUT * 2, 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int g = i * 64 + lane, pix = g >> 2, c = g & 3;                // 32 pixels x 4 chunks of 8 channels
        const int pg = pg0 + pix;
        if (pg < npx) {
            int row = (int)(((float)pg + 0.5f) * inv_w2), col = pg - row * w2;
            if (col < 0) { --row; col += w2; } else if (col >= w2) { ++row; col -= w2; }
            const int f = (pix >> 1) & 3;
            const u4v v = stage[pix * CR + (c ^ f)];
            const int off = ((m_row0 + row * OW + col) * p.COUT + 8 * c) * 2;
            __builtin_amdgcn_raw_buffer_store_b128(v, rout, off, 0, 0);
        }
    }
}

// conv1 (uint8 frame in, kPf = 3 trips = the whole K of 5 kernel rows x 16 bytes): resident weights, persistent workgroups,
// independent waves, no barrier after the weight stage.  A wave walks 32-pixel tiles; the raw dwords of the NEXT tile are
// requested before the current tile's MFMAs and epilogue, so their latency hides behind them.
__global__ __launch_bounds__(1024) void trs_conv_u8_kernel(const ConvParams p)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    constexpr int NBW = 32, kGr = 6;                                        // granules per lane: 3 trips x 2 k-steps
    u4v* lw = reinterpret_cast<u4v*>(psmem);                               // [G_pad][NBW] granules
    int* lgoff = reinterpret_cast<int*>(psmem + (size_t)p.G_pad * NBW * 16);
    const size_t off_bias = (size_t)p.G_pad * NBW * 16 + (((size_t)p.G_pad * 4 + 15) & ~(size_t)15);
    float4* lbias = reinterpret_cast<float4*>(psmem + off_bias);           // [NBW / 4]
    u4v* stage = reinterpret_cast<u4v*>(psmem + off_bias + NBW * 4) + wave * 128;   // 2 KB per wave: output transpose
    for (int i = tid; i < NBW / 4; i += blockDim.x) lbias[i] = *reinterpret_cast<const float4*>(p.bias + 4 * i);
    for (int i = tid; i < p.G_pad * NBW; i += blockDim.x) {
        const int g = i / NBW, c = i - g * NBW;
        lw[i] = p.w[(size_t)g * p.COUT_PAD + c];
    }
    for (int i = tid; i < p.G_pad; i += blockDim.x) lgoff[i] = p.goff[i];
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
    const int ohw = p.OH * p.OW;
    const int ntiles = (p.M + 31) >> 5, stride = gridDim.x * nwaves;
    const float inv_ow = 1.0f / (float)p.OW;
    int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * nwaves + wave);
    if (tile >= ntiles) return;
    int goffs[kGr];                                                         // this lane's granules: k-step s, half h -> granule 2s + h
#pragma unroll
    for (int s = 0; s < kGr; ++s) goffs[s] = lgoff[min(2 * s + h, p.G_pad - 1)];
    auto base_of = [&](int t) {                                             // t is wave-uniform: the frame split runs on the scalar unit
        const int n0 = (t * 32) / ohw, rem0 = t * 32 - n0 * ohw;
        return window_base(p, n0, rem0, min(r, p.M - 1 - t * 32), inv_ow);
    };
    unsigned raw[kGr][3];
    int pixbase = base_of(tile);
    auto request = [&](int pb) {
#pragma unroll
        for (int s = 0; s < kGr; ++s) {
            const int al = (pb + goffs[s]) & ~3;
            raw[s][0] = __builtin_amdgcn_raw_buffer_load_b32(rin, al, 0, 0);
            raw[s][1] = __builtin_amdgcn_raw_buffer_load_b32(rin, al + 4, 0, 0);
            raw[s][2] = __builtin_amdgcn_raw_buffer_load_b32(rin, al + 8, 0, 0);
        }
    };
    request(pixbase);
    while (true) {
        // 8 of the 16 bytes of a kernel row at any byte alignment: byte-align, then exact u8 -> binary16 (0..255 is exact)
        h16x8 x[kGr];
#pragma unroll
        for (int s = 0; s < kGr; ++s) {
            const unsigned sh = (unsigned)(pixbase + goffs[s]) & 3u;
            const unsigned lo = __builtin_amdgcn_alignbyte(raw[s][1], raw[s][0], sh);
            const unsigned hi = __builtin_amdgcn_alignbyte(raw[s][2], raw[s][1], sh);
            auto pair = [](unsigned w, int j) -> unsigned {
                const float f0 = (float)((w >> (8 * j)) & 255u), f1 = (float)((w >> (8 * j + 8)) & 255u);
                return u8pair_h16(f0, f1);
            };
            const u4v packed = {pair(lo, 0), pair(lo, 2), pair(hi, 0), pair(hi, 2)};
            x[s] = __builtin_bit_cast(h16x8, packed);
        }
        const int next = tile + stride;                                     // uniform
        const int nbase = base_of(min(next, ntiles - 1));
        request(nbase);                                                     // unconditional: the last tile re-requests itself
        f32x16 acc[1];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[0][i] = 0.0f;
#pragma unroll
        for (int s = 0; s < kGr; ++s) {
            if (2 * s < p.G_pad) {
                const h16x8 w = __builtin_bit_cast(h16x8, lw[(2 * s + h) * NBW + r]);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x[s], acc[0], 0, 0, 0);
            }
        }
        store_tile<1>(stage, acc, lbias, p, tile, 0, lane);
        if (next >= ntiles) break;
        tile = next; pixbase = nbase;
    }
}

// conv2..conv7 (fp16 input): resident weights + QUAD-COALESCED pixel loads.  Counters showed the per-lane 16-byte loads of
// the kernel above cost the texture addresser one tag lookup per lane (56-88 per instruction: every lane another line) and
// bound every layer.  Here the four lanes of a quad fetch the four consecutive granules (64 contiguous bytes) of ONE pixel,
// so an instruction is 16 pixels x 64 B; the fragments reach the MFMA layout through a 2 KB wave-private LDS stage:
//   load  (trip T, instruction i): lane l = 4q + jj holds granule 4T + j of pixel 16i + q, j = (jj - (q >> 2)) & 3
//   write : lane l -> stage[i][l]                                  (linear, conflict-free)
//   read  (k-step s): lane (r, h) wants granule j = 2s + h of pixel r  ->  stage[r >> 4][4q + ((j + (q >> 2)) & 3)], q = r & 15
// The rotation by q >> 2 spreads the 16 lanes of every ds_read_b128 group over all 16 slots of the 256-B bank row.
// Granules of a trip are contiguous in memory: k is ordered (kh, [kw, cin]) and a kernel row is one contiguous run of
// KW * CIN / 8 granules in NHWC, padded to a multiple of 4 (only conv2: 15 -> 16, zero weights).
template <int NB>
__global__ __launch_bounds__(1024) void trs_conv_lt_kernel(const ConvParams p)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    constexpr int NBW = NB * 32;
    const int cbase = blockIdx.y * NBW;
    u4v* lw = reinterpret_cast<u4v*>(psmem);                               // [G_pad][NBW] granules
    const size_t off_goff = (size_t)p.G_pad * NBW * 16;
    const size_t off_bias = off_goff + (((size_t)p.G_pad * 4 + 15) & ~(size_t)15);
    const size_t off_stage = off_bias + NBW * 4;
    int* lgoff = reinterpret_cast<int*>(psmem + off_goff);
    float4* lbias = reinterpret_cast<float4*>(psmem + off_bias);
    u4v* stage = reinterpret_cast<u4v*>(psmem + off_stage) + wave * 128;   // [2][64] granules of this wave
    for (int i = tid; i < NBW / 4; i += blockDim.x) lbias[i] = *reinterpret_cast<const float4*>(p.bias + cbase + 4 * i);
    for (int i = tid; i < p.G_pad * NBW; i += blockDim.x) {
        const int g = i / NBW, c = i - g * NBW;
        lw[i] = p.w[(size_t)g * p.COUT_PAD + cbase + c];
    }
    for (int i = tid; i < p.G_pad; i += blockDim.x) lgoff[i] = p.goff[i];
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
    const int ohw = p.OH * p.OW;
    const int ntiles = (p.M + 31) >> 5;
    constexpr int kPf = kConvPrefetch;
    // loader role of this lane: pixel 16 i + lq of the tile, granule lj of every trip
    const int lq = lane >> 2, lj = ((lane & 3) - (lq >> 2)) & 3;
    // reader role: slots of k-step 0 and 1 for pixel r, half h
    const int rq = r & 15;
    const int rd0 = (r >> 4) * 64 + 4 * rq + ((h + (rq >> 2)) & 3);
    const int rd1 = (r >> 4) * 64 + 4 * rq + ((2 + h + (rq >> 2)) & 3);
    const float inv_ow = 1.0f / (float)p.OW;
    const int stride = gridDim.x * nwaves;
    int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * nwaves + wave);
    if (tile >= ntiles) return;
    int lbase[2];
    auto bases_of = [&](int t, int (&out)[2]) {                             // t is wave-uniform: the frame split runs on the scalar unit
        const int n0 = (t * 32) / ohw, rem0 = t * 32 - n0 * ohw;
#pragma unroll
        for (int i = 0; i < 2; ++i) out[i] = window_base(p, n0, rem0, min(16 * i + lq, p.M - 1 - t * 32), inv_ow);
    };
    auto load_q = [&](int g4, int i) -> u4v {                               // g4 = first granule of the trip
        return __builtin_amdgcn_raw_buffer_load_b128(rin, lbase[i] + lgoff[g4 + lj], 0, 0);
    };
    u4v ring[2 * kPf];
    auto preload = [&]() {
#pragma unroll
        for (int t = 0; t < kPf; ++t) {
            const int g = min(4 * t, p.G_pad - 4);
            ring[2 * t] = load_q(g, 0); ring[2 * t + 1] = load_q(g, 1);
        }
    };
    bases_of(tile, lbase);
    preload();
    while (true) {
        f32x16 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nb][i] = 0.0f;
        auto trip_mfma = [&](int gt, int t) {
#if TRS_CONV_ABLATE == 2
            const h16x8 x0 = __builtin_bit_cast(h16x8, ring[2 * t]);
            const h16x8 x1 = __builtin_bit_cast(h16x8, ring[2 * t + 1]);
#else
            stage[lane] = ring[2 * t];                                      // transpose through the wave's LDS stage (in-order per wave)
            stage[64 + lane] = ring[2 * t + 1];
            const h16x8 x0 = __builtin_bit_cast(h16x8, stage[rd0]);
            const h16x8 x1 = __builtin_bit_cast(h16x8, stage[rd1]);
#endif
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const h16x8 w = __builtin_bit_cast(h16x8, lw[(gt + h) * NBW + nb * 32 + r]);
#if TRS_CONV_ABLATE == 3
                acc[nb][0] += (float)w[0] * (float)x0[0];
#else
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x0, acc[nb], 0, 0, 0);
#endif
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const h16x8 w = __builtin_bit_cast(h16x8, lw[(gt + 2 + h) * NBW + nb * 32 + r]);
#if TRS_CONV_ABLATE == 3
                acc[nb][1] += (float)w[0] * (float)x1[0];
#else
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x1, acc[nb], 0, 0, 0);
#endif
            }
        };
        int g2 = 0;
        for (; g2 + 4 * kPf < p.G_pad; g2 += 4 * kPf) {
#pragma unroll
            for (int t = 0; t < kPf; ++t) {
                trip_mfma(g2 + 4 * t, t);
#if TRS_CONV_ABLATE != 1
                const int gn = min(g2 + 4 * t + 4 * kPf, p.G_pad - 4);
                ring[2 * t] = load_q(gn, 0); ring[2 * t + 1] = load_q(gn, 1);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int t = 0; t < kPf; ++t)
            if (g2 + 4 * t < p.G_pad) trip_mfma(g2 + 4 * t, t);

        // the next tile's first trips are requested before this tile's epilogue (unconditional: the last tile re-requests itself)
        const int next = tile + stride;
        bases_of(min(next, ntiles - 1), lbase);
        preload();
#if TRS_CONV_ABLATE != 4
        store_tile<NB>(stage, acc, lbias, p, tile, cbase, lane);
#endif
        if (next >= ntiles) break;
        tile = next;
    }
}

// conv4 .. conv7 (3x3, stride 1): the whole input activation of a frame is 13-26 KB, so F frames of it live in LDS and the nine
// overlapping windows of every output pixel are read from there — the quad-load kernel above fetched every input byte nine times
// through the texture addresser.  One workgroup = F frames:
//   staging   input granules (8 channels = 16 B) global -> LDS by LDS-DMA, XOR-swizzled within the pixel's row of granules
//             (physical slot q = g ^ swz(pixel); the swizzle goes on the SOURCE address, the LDS side stays lane-linear) so that
//             the 16 lanes of a ds_read_b128 group — 16 consecutive pixels, one logical granule — cover all 64 banks
//   work item a super-tile of NT x 32 output pixels (consecutive over the workgroup's frames) x NB x 32 output channels, one wave;
//             per k-step (16 input channels of one tap): NT ds_read_b128 (pixels, B operand), NB weight granules straight from
//             L2 (A operand, 512 contiguous bytes per half wave, prefetched kFrameRing k-steps ahead in registers), NT x NB MFMAs.
//             Each pixel fragment feeds NB MFMAs and each weight fragment NT: LDS and L2 each supply half of what one-to-one
//             feeding would need (LDS 128 B/clk and L2 64 B/clk per CU are what bound a 32x32x16 MFMA stream otherwise)
//   epilogue  bias + ReLU + fp16, 8-byte stores (these activations are small: 9-19 KB per frame)
// conv2 / conv3 (stride-2 5x5): overlapping windows make the kernel above fetch every input byte ~2.5x, and the texture
// addresser (about one lookup per clock) is what bounds these layers.  Here a wave stages, per kernel row, the CONTIGUOUS
// input span its 32-pixel tile needs (a tile crosses output rows, so the span is 1..4 segments, one per output row touched)
// with fully coalesced 1 KB loads, writes it to a wave-private LDS stage, and reads the im2col fragments from there:
//   virtual granule v of the stage = segment start c_s + (input granule within the segment's span)
//   loader lane l, instruction k: v = 64 k + l  ->  global address base_s + kh * row_bytes + 16 (v - c_s)
//   reader lane (pixel r = segment s, position q; half h), k-step t: v = c_s + q * S * cg + 2 t + h
// The next kernel row's span is requested (registers) before the current row's MFMAs; SWZ (pixel stride of 8 granules,
// conv3) XOR-swizzles the stage so a ds_read_b128 group covers all 16 slots of the bank row.
template <int NB, bool SWZ>
__global__ __launch_bounds__(768) void trs_conv_span_kernel(const ConvParams p)
{
    constexpr int kMaxNl = 5, kMaxSeg = 4;                                  // 12 waves per workgroup at most: the segment bookkeeping wants ~150 VGPRs
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    constexpr int NBW = NB * 32;
    const int cbase = blockIdx.y * NBW;
    u4v* lw = reinterpret_cast<u4v*>(psmem);                               // [G_pad][NBW] granules
    const size_t off_bias = (size_t)p.G_pad * NBW * 16;
    const size_t off_stage = off_bias + NBW * 4;
    float4* lbias = reinterpret_cast<float4*>(psmem + off_bias);
    u4v* stage = reinterpret_cast<u4v*>(psmem + off_stage) + wave * (p.span_nl * 64);   // span_nl KB per wave
    for (int i = tid; i < NBW / 4; i += blockDim.x) lbias[i] = *reinterpret_cast<const float4*>(p.bias + cbase + 4 * i);
    for (int i = tid; i < p.G_pad * NBW; i += blockDim.x) {
        const int g = i / NBW, c = i - g * NBW;
        lw[i] = p.w[(size_t)g * p.COUT_PAD + cbase + c];
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.M * p.COUT * 2, 0x00020000);
    const int ohw = p.OH * p.OW;
    const int ntiles = (p.M + 31) >> 5, stride = gridDim.x * nwaves;
    const int row_bytes = p.IW * p.in_px_bytes;
    const int pix_gran = p.S * p.cg;                                        // granules between neighbouring output pixels
    const int tail = p.run_pad;                                             // granules a segment's last pixel needs (window + run padding)
    auto swz = [](int v) { return SWZ ? v ^ ((v >> 4) & 7) : v; };

    int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * nwaves + wave);
    if (tile >= ntiles) return;
    int laddr[kMaxNl];                                                      // loader: byte address of this lane's granule of instruction k, kernel row 0
    auto setup = [&](int t, int& roff) {                                    // roff: virtual granule of this lane's pixel, k-step 0, half h
        // t is wave-uniform: the segment list is scalar work
        const int m0 = t * 32;
        const int n0 = m0 / ohw, rem0 = m0 - n0 * ohw;
        int n = n0, oy = rem0 / p.OW, ox = rem0 - oy * p.OW;
        int left = 32, cum = 0, c = 0;
        roff = 0;
#pragma unroll
        for (int k = 0; k < kMaxNl; ++k) laddr[k] = p.in_bytes;             // out of range: the buffer load returns zeros
#pragma unroll
        for (int sgi = 0; sgi < kMaxSeg; ++sgi) {
            if (left > 0) {
                const int len = min(left, p.OW - ox);
                const int sg = (len - 1) * pix_gran + tail;                 // granules of this segment's span
                const int base = ((n * p.IH + oy * p.S) * p.IW + ox * p.S) * p.in_px_bytes;
                if (r >= cum && r < cum + len) roff = c + (r - cum) * pix_gran + h;
#pragma unroll
                for (int k = 0; k < kMaxNl; ++k) {
                    const int v = 64 * k + lane;
                    if (k < p.span_nl && v >= c && v < c + sg) laddr[k] = base + (v - c) * 16;
                }
                left -= len; cum += len; c += sg;
                ox = 0; ++oy;
                if (oy == p.OH) { oy = 0; ++n; }
            }
        }
    };
    u4v regs[kMaxNl];
    auto request = [&](int kh) {
#pragma unroll
        for (int k = 0; k < kMaxNl; ++k)
            if (k < p.span_nl) regs[k] = __builtin_amdgcn_raw_buffer_load_b128(rin, laddr[k] == p.in_bytes ? p.in_bytes : laddr[k] + kh * row_bytes, 0, 0);
    };
    int roff_cur = 0, roff_next = 0;
    setup(tile, roff_cur);
    request(0);
    while (true) {
        f32x16 acc[NB];
        acc_bias<NB>(acc, lbias, h);                                        // the sums start at the bias
        const int next = tile + stride;                                     // uniform
        for (int kh = 0; kh < p.KH; ++kh) {
#pragma unroll
            for (int k = 0; k < kMaxNl; ++k)
                if (k < p.span_nl) stage[swz(64 * k + lane)] = regs[k];     // this kernel row's span -> LDS (in-order per wave)
            if (kh + 1 < p.KH) request(kh + 1);                             // uniform branch; next row's span flies during the MFMAs
            else { setup(min(next, ntiles - 1), roff_next); request(0); }   // ... or the next tile's first row (the last tile re-requests itself)
            const int gbase = kh * p.run_pad;
            for (int t = 0; t < p.run_pad; t += 2) {                        // (fetching step t + 1's fragments by hand before step t's MFMAs measured 5-10 % slower)
                const h16x8 x = __builtin_bit_cast(h16x8, stage[swz(roff_cur + t)]);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const h16x8 w = __builtin_bit_cast(h16x8, lw[(gbase + t + h) * NBW + nb * 32 + r]);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, acc[nb], 0, 0, 0);
                }
            }
        }
        store_tile_direct<NB>(acc, p, rout, tile, cbase, lane);
        if (next >= ntiles) break;
        tile = next; roff_cur = roff_next;
    }
}

