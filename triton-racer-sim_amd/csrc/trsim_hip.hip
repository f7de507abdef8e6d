// trsim_hip.hip — libtrsim.so: the gfx950 (MI355X, CDNA4) kernels and the C ABI of include/trsim.h.
//
// One fused kernel per env step (trs_step_kernel).  A workgroup (960 threads = 15 wave64) owns a
// contiguous range of envs and keeps every read-only table in LDS for the whole launch:
//
//   LDS image (copied once per launch from one contiguous device blob, 16 B per lane):
//     px[np] py[np] pz[np]  binary64 raw track points (SoA)      — nearest-point search operands
//     map[map_h][map_words] packed 2-bit surface classes          — rasteriser lookup
//     rowtab[H] (row_lz,row_k) float2, palette[H][4] 0x00BBGGRR  — camera rows
//     + a small scratch area (query points, per-wave argmin partials, camera params)
//
//   per step, per chunk of <=16 envs of the workgroup:
//     phase 0  lane j of wave 0 integrates env j (bicycle model, include/trsim_spec.h) — coalesced SoA loads
//     phase A  all 15 waves scan the LDS track for each env: binary64 L1 distance, per-lane strict '<',
//              wave64 butterfly argmin (lowest index wins ties) -> one partial per wave
//              (= reference LocationTracker.__find_closest, components/track_data_process.py:89-101)
//     phase A2 lane j of wave 0 folds env j's 15 partials, computes y, cte, done, reward, stores the SoA
//              state (coalesced) and the env's camera parameters; off-track envs are counted with a
//              wave ballot + one atomic
//     phase B  all threads rasterise the chunk: a lane produces 4 consecutive pixels (12 B) of the
//              flattened HxWx3 stream, so one wave-instruction stores 768 contiguous bytes (6 full 128-B lines)
//
// Bound: HBM writes of the image (57,600 B per env-step at 120x160) — see DESIGN.md.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (spec rule R1: no implicit FMA contraction).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/trsim.h"
#include "../../include/trsim_spec.h"
#include "trsim_tables.hpp"

#ifndef TRS_ABLATE
#define TRS_ABLATE 0   /* 0 = product; 1/2 = timing-only diagnostic builds (scripts/ablate.sh), never shipped */
#endif

#define TRS_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

constexpr int kBlock = 960;            // 15 waves: 4800 four-pixel groups of a 120x160 image = 5 x 960
constexpr int kWaves = kBlock / 64;
constexpr int kEMax = 15;              // envs of one workgroup processed per chunk (<= one wave each in the search)
constexpr int kLocBlock = 1024;        // locate kernel: 16 waves = 16 queries in flight per workgroup

struct KParams {
    // per-env state, SoA
    float *x, *y, *z, *yaw, *v, *speed, *cte, *ep_return, *last_return, *steer_filt;
    int32_t *seg_idx, *ep_len;
    uint8_t *done, *pending;
    const float *ctl_steer, *ctl_thr, *ctl_brk;
    const uint8_t* ctl_reset;
    uint8_t* img[2];
    const unsigned char* blob;
    const float* start_yaw;            // [np]
    const float* tangent_g;            // [np][2] global copy, used when the table does not fit in LDS (tan_in_lds == 0)
    unsigned long long* stats;         // [0] off-track events, [1] resets, [2] layout-assumption failures
    int n_envs, env_id_base, envs_per_wg;
    int np, H, W, gpr, gpe;            // groups (4 px) per row / per env
    unsigned row_magic;                // q / gpr == umulhi(q, row_magic) for q < gpe (checked on the host)
    int map_w, map_h, map_words, map_pitch_b;   // map_pitch_b: bytes per map row in the LDS image (odd number of words)
    int off_map, off_px, off_py, off_pz, off_tan, off_rowtab, off_pal, blob_bytes, off_scratch;   // LDS image: map at 0
    float map_x0f, map_z0f, inv_cellf;
    float dt, max_steer, inv_wheelbase, accel_max, drag_lin, roll_res, brake_max;
    float v_max, v_rev_max, offtrack_cte, offtrack_penalty, cam_fwd;
    int auto_reset, render, synth, n_steps, img_parity, stage_bytes, rows_per_pass, tan_in_lds;
    uint32_t step0;
    unsigned long long seed;
};

// ---------------------------------------------------------------------------------------------
// device pieces of the spec

__device__ __forceinline__ void spec_sincos(float a, float& so, float& co)
{
    const float q = rintf(a * TRS_TWO_OVER_PI);
    float r = fmaf(q, -TRS_PIO2_HI, a);
    r = fmaf(q, -TRS_PIO2_LO, r);
    const float zz = r * r;
    const float ps = fmaf(fmaf(TRS_S0, zz, TRS_S1), zz, TRS_S2);
    const float s = fmaf(r * zz, ps, r);
    const float pc = fmaf(fmaf(TRS_C0, zz, TRS_C1), zz, TRS_C2);
    const float c = fmaf(zz * zz, pc, fmaf(zz, -0.5f, 1.0f));
    const int n = ((int)q) & 3;
    so = (n == 0) ? s : (n == 1) ? c : (n == 2) ? -s : -c;
    co = (n == 0) ? c : (n == 1) ? -s : (n == 2) ? -c : s;
}

__device__ __forceinline__ float clampf(float a, float lo, float hi) { return a < lo ? lo : (a > hi ? hi : a); }

// wave64 argmin over (distance, index): smaller distance wins, equal distance -> lower index.
// DPP reduction (row_shr 1,2,4,8 then row_bcast15 / row_bcast31): data moves through the VALU's DPP path
// instead of the LDS crossbar that __shfl (ds_bpermute) uses.  The wave's result ends in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void argmin_dpp_step(double& d, int& i)
{
    const int lo = __double2loint(d), hi = __double2hiint(d);
    // lanes without a valid source (row edge / masked rows) read their own value: combining with self is a no-op
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    const int oi = __builtin_amdgcn_update_dpp(i, i, CTRL, ROW_MASK, 0xf, false);
    const double od = __hiloint2double(ohi, olo);
    const bool take = (od < d) || (od == d && oi < i);
    d = take ? od : d;
    i = take ? oi : i;
}

__device__ __forceinline__ void wave_argmin(double& d, int& i)
{
    argmin_dpp_step<0x111, 0xf>(d, i);   // row_shr:1
    argmin_dpp_step<0x112, 0xf>(d, i);   // row_shr:2
    argmin_dpp_step<0x114, 0xf>(d, i);   // row_shr:4
    argmin_dpp_step<0x118, 0xf>(d, i);   // row_shr:8   -> lane 15 of each row holds the row's result
    argmin_dpp_step<0x142, 0xa>(d, i);   // row_bcast:15 into rows 1 and 3
    argmin_dpp_step<0x143, 0xc>(d, i);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's result
}

__device__ __forceinline__ unsigned cvt_u32_sat(float x)
{
    // v_cvt_u32_f32: truncate toward zero, saturate (negative / NaN -> 0).  For the rasteriser's
    // clamp(floor(g), 0, G-1) this equals min(cvt_u32_sat(g), G-1): g < 0 -> 0, g >= 0 -> trunc == floor.
    unsigned r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

__device__ __forceinline__ void synth_controls(unsigned long long seed, uint32_t gid, uint32_t step, float& sf, float& steer, float& thr)
{
    unsigned long long z = seed + (((unsigned long long)gid << 32) | (unsigned long long)step) * 0x9E3779B97F4A7C15ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float us = (float)(uint32_t)(z >> 40) * 5.9604644775390625e-08f;
    const float ut = (float)(uint32_t)((z >> 16) & 0xFFFFFFu) * 5.9604644775390625e-08f;
    const float raw = us * 2.0f - 1.0f;
    sf = sf + TRS_SYNTH_ALPHA * (raw - sf);
    steer = sf;
    thr = TRS_SYNTH_THR_LO + TRS_SYNTH_THR_SPAN * ut;
}

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));   // 16-B register tuple (HIP's uint4 struct defeats SROA here)
typedef __attribute__((address_space(3))) const uint32_t* lds_u32p;

constexpr int kStagers = kBlock - 64;  // waves 1..14 stage tables while wave 0 integrates
constexpr int kHotRegs = 7;            // 7 x 896 x 16 B = 100 KB >= points + tangents + camera tables
constexpr int kMapRegs = 7;            // >= TRS_MAP_LDS_BUDGET (96 KB) of packed map

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void trs_step_kernel(const KParams p)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;        // LDS byte address of the dynamic segment
    if (lds0 + (unsigned)p.off_map != 0u) {                 // the rasteriser's map addressing assumes LDS offset 0
        if (tid == 0 && blockIdx.x == 0) atomicAdd(&p.stats[2], 1ull);
        return;
    }

    const double* lpx = reinterpret_cast<const double*>(smem + p.off_px);
    const double* lpy = reinterpret_cast<const double*>(smem + p.off_py);
    const double* lpz = reinterpret_cast<const double*>(smem + p.off_pz);
    const float2* ltan = reinterpret_cast<const float2*>(smem + p.off_tan);
    const f2v* lrow = reinterpret_cast<const f2v*>(smem + p.off_rowtab);
    // scratch (16-B aligned base): camera params first (float4), then binary64 arrays, then ints
    float4* scam = reinterpret_cast<float4*>(smem + p.off_scratch);               // [kEMax] camx, camz, s, c
    double* sq = reinterpret_cast<double*>(scam + kEMax);                         // [3][kEMax] query points
    double* spd = sq + 3 * kEMax;                                                 // [kEMax][kWaves] partial distance
    int* spi = reinterpret_cast<int*>(spd + kEMax * kWaves);                      // [kEMax][kWaves] partial index
    const double* gpx = reinterpret_cast<const double*>(p.blob + p.off_px);       // global copies (reset path only)
    const double* gpy = reinterpret_cast<const double*>(p.blob + p.off_py);
    const double* gpz = reinterpret_cast<const double*>(p.blob + p.off_pz);

    // ---- prologue: waves 1..14 stage the read-only tables (register-staged, asynchronous) ----
    //   hot region  [off_px, off_scratch): points, tangents, camera rows, palette -> written to LDS right away
    //   map region  [0, off_px): held in registers and written to LDS just before the first raster phase, so the
    //   88 KB of L2->LDS traffic overlaps phase 0 / A / A2 of the first step (plain loads survive s_barrier)
    u4v mreg[kMapRegs];
#pragma unroll
    for (int r = 0; r < kMapRegs; ++r) mreg[r] = (u4v)(0u);
    bool map_pending = false;
    if (wave != 0) {
        const int st = tid - 64;
        const u4v* hsrc = reinterpret_cast<const u4v*>(p.blob + p.off_px);
        u4v* hdst = reinterpret_cast<u4v*>(smem + p.off_px);
        const int hot16 = (p.off_scratch - p.off_px) >> 4;
        u4v hreg[kHotRegs];
#pragma unroll
        for (int r = 0; r < kHotRegs; ++r) { const int i = st + r * kStagers; hreg[r] = (i < hot16) ? hsrc[i] : (u4v)(0u); }
        if (p.render) {
            const u4v* msrc = reinterpret_cast<const u4v*>(p.blob);
            const int map16 = p.off_px >> 4;
#pragma unroll
            for (int r = 0; r < kMapRegs; ++r) { const int i = st + r * kStagers; if (i < map16) mreg[r] = msrc[i]; }
            map_pending = true;
        }
#pragma unroll
        for (int r = 0; r < kHotRegs; ++r) { const int i = st + r * kStagers; if (i < hot16) hdst[i] = hreg[r]; }
    }

    const int e_begin = blockIdx.x * p.envs_per_wg;
    const int e_end = min(e_begin + p.envs_per_wg, p.n_envs);

    for (int k = 0; k < p.n_steps; ++k) {
        const uint32_t t = p.step0 + (uint32_t)k;
        uint8_t* const img = p.img[(p.img_parity + k) & 1];

        for (int c0 = e_begin; c0 < e_end; c0 += kEMax) {
            const int nE = min(kEMax, e_end - c0);
            const int e = c0 + tid;          // env of this lane in phases 0 / A2 (tid < nE)

            // ---- phase 0: integrate (lane j of wave 0 <-> env j of the chunk: coalesced SoA accesses) ----
            float x1 = 0.f, y0 = 0.f, z1 = 0.f, yaw1 = 0.f, v2 = 0.f, hs = 0.f, hc = 1.f, sf = 0.f;
            int do_reset = 0, prev_idx = 0;
            if (tid < nE) {
                const int gid = p.env_id_base + e;
                const bool pend = p.pending[e] != 0;
                const bool was_done = p.done[e] != 0;
                const bool reset_in = (!p.synth && p.ctl_reset && k == 0) ? (p.ctl_reset[e] != 0) : false;
                do_reset = pend || reset_in || (p.auto_reset && was_done);
                sf = p.steer_filt[e];
                prev_idx = p.seg_idx[e];
                if (do_reset) {
                    const int si = (int)(((long long)TRS_START_STRIDE * gid) % p.np);
                    x1 = (float)gpx[si]; y0 = (float)gpy[si]; z1 = (float)gpz[si];
                    yaw1 = p.start_yaw[si]; v2 = 0.0f; sf = 0.0f;
                    spec_sincos(yaw1, hs, hc);
                } else {
                    float steer, thr, brk = 0.0f;
                    if (p.synth) synth_controls(p.seed, (uint32_t)gid, t, sf, steer, thr);
                    else { steer = p.ctl_steer[e]; thr = p.ctl_thr[e]; brk = p.ctl_brk ? p.ctl_brk[e] : 0.0f; }
                    steer = clampf(steer, -1.0f, 1.0f);
                    thr = clampf(thr, -1.0f, 1.0f);
                    brk = clampf(brk, 0.0f, 1.0f);
                    float sd, cd;
                    spec_sincos(steer * p.max_steer, sd, cd);
                    const float tan_d = sd / cd;
                    const float v = p.v[e];
                    const float a = thr * p.accel_max - p.drag_lin * v;
                    const float v1 = v + a * p.dt;
                    const float dv = (p.roll_res + brk * p.brake_max) * p.dt;
                    if (v1 > 0.0f) { v2 = v1 - dv; if (v2 < 0.0f) v2 = 0.0f; }
                    else if (v1 < 0.0f) { v2 = v1 + dv; if (v2 > 0.0f) v2 = 0.0f; }
                    else v2 = 0.0f;
                    v2 = clampf(v2, -p.v_rev_max, p.v_max);
                    yaw1 = p.yaw[e] + ((v2 * tan_d) * p.inv_wheelbase) * p.dt;
                    if (yaw1 > TRS_PI) yaw1 -= TRS_TWO_PI;
                    if (yaw1 < -TRS_PI) yaw1 += TRS_TWO_PI;
                    spec_sincos(yaw1, hs, hc);
                    x1 = p.x[e] + (v2 * hs) * p.dt;
                    z1 = p.z[e] + (v2 * hc) * p.dt;
                    y0 = p.y[e];
                }
                sq[tid] = (double)x1; sq[kEMax + tid] = (double)y0; sq[2 * kEMax + tid] = (double)z1;
            }
            __syncthreads();

            // ---- phase A: nearest raw track point, L1 in binary64 ----
            // the 15 waves are dealt to the chunk's envs: env j is scanned by waves [j*wpe, (j+1)*wpe), each taking
            // an interleaved slice of the points; one DPP argmin per wave, one partial per (env, slice)
            const int wpe = kWaves / nE;
            {
                const int j = wave / wpe, slice = wave - j * wpe;
                if (j < nE) {
                    const double qx = sq[j], qy = sq[kEMax + j], qz = sq[2 * kEMax + j];
                    double best = TRS_LOST_L1;
                    int bi = 0;
                    for (int i = slice * 64 + lane; i < p.np; i += wpe * 64) {
                        const double d = (fabs(qx - lpx[i]) + fabs(qy - lpy[i])) + fabs(qz - lpz[i]);
                        if (d < best) { best = d; bi = i; }
                    }
                    wave_argmin(best, bi);
                    if (lane == 63) { spd[j * kWaves + slice] = best; spi[j * kWaves + slice] = bi; }
                }
            }
            __syncthreads();

            // ---- phase A2: fold partials, finish the env, store state ----
            int is_done = 0;
            if (tid < nE) {
                double best = spd[tid * kWaves];
                int idx = spi[tid * kWaves];
                for (int w = 1; w < wpe; ++w) {
                    const double od = spd[tid * kWaves + w];
                    const int oi = spi[tid * kWaves + w];
                    const bool take = (od < best) || (od == best && oi < idx);
                    best = take ? od : best;
                    idx = take ? oi : idx;
                }
                const float y1 = (float)lpy[idx];
                const float2 tg = p.tan_in_lds ? ltan[idx] : reinterpret_cast<const float2*>(p.tangent_g)[idx];
                const float cte = (x1 - (float)lpx[idx]) * tg.y - (z1 - (float)lpz[idx]) * tg.x;
                const bool lost = best >= TRS_LOST_L1;
                is_done = (fabsf(cte) > p.offtrack_cte) || lost;
                float epr = p.ep_return[e];
                int epl = p.ep_len[e];
                if (do_reset) {
                    p.last_return[e] = epr;
                    epr = 0.0f; epl = 0;
                    p.pending[e] = 0;
                } else {
                    int d = idx - prev_idx;
                    const int half = p.np / 2;
                    if (d >= p.np - half) d -= p.np;
                    if (d < -half) d += p.np;
                    const float reward = (float)d - (is_done ? p.offtrack_penalty : 0.0f);
                    epr = epr + reward;
                    epl += 1;
                }
                p.x[e] = x1; p.y[e] = y1; p.z[e] = z1; p.yaw[e] = yaw1; p.v[e] = v2;
                p.speed[e] = fabsf(v2); p.cte[e] = cte; p.seg_idx[e] = idx; p.done[e] = (uint8_t)is_done;
                p.ep_return[e] = epr; p.ep_len[e] = epl; p.steer_filt[e] = sf;
                const float camx = ((x1 + p.cam_fwd * hs) - p.map_x0f) * p.inv_cellf;
                const float camz = ((z1 + p.cam_fwd * hc) - p.map_z0f) * p.inv_cellf;
                scam[tid] = make_float4(camx, camz, hs, hc);
            }
            if (wave == 0) {   // off-track / reset census: wave ballot, one atomic per workgroup-chunk
                const unsigned long long mdone = __ballot(is_done != 0);
                const unsigned long long mreset = __ballot(do_reset != 0);
                if (lane == 0) {
                    if (mdone) atomicAdd(&p.stats[0], (unsigned long long)__popcll(mdone));
                    if (mreset) atomicAdd(&p.stats[1], (unsigned long long)__popcll(mreset));
                }
            }
            if (map_pending) {   // first raster of this launch: land the map slice held in registers
                u4v* mdst = reinterpret_cast<u4v*>(smem);
                const int map16 = p.off_px >> 4;
                const int st = tid - 64;
#pragma unroll
                for (int r = 0; r < kMapRegs; ++r) { const int i = st + r * kStagers; if (i < map16) mdst[i] = mreg[r]; }
                map_pending = false;
            }
            __syncthreads();

            // ---- phase B: rasterise the chunk ----
            // A thread owns one 4-pixel column group (u0 fixed) and walks image rows, so the pixel-centre offsets
            // uf are loop constants.  Per pixel: 1 packed fma (gx,gz), 2 saturating converts + 2 min (= floor + clamp),
            // 3 address ops, 1 LDS map read, shift + bit-field extract, 1 palette address op, 1 LDS palette read.
            if (p.render) {
                const float half_w = (float)(p.W / 2);
                const unsigned gwm1 = (unsigned)(p.map_w - 1), ghm1 = (unsigned)(p.map_h - 1);
                const int cg = tid % p.gpr, r0 = tid / p.gpr;       // threads with r0 >= rows_per_pass idle (none at W = 160)
                const float uf0 = (float)(cg << 2) + 0.5f - half_w;
                const f2v ufa = {uf0, uf0}, ufb = {uf0 + 1.0f, uf0 + 1.0f}, ufc = {uf0 + 2.0f, uf0 + 2.0f}, ufd = {uf0 + 3.0f, uf0 + 3.0f};
                const unsigned pitch = (unsigned)p.map_pitch_b;
                const int vstart = r0 < p.rows_per_pass ? r0 : p.H;
                const size_t row_bytes = (size_t)p.gpr * 12;
                for (int j = 0; j < nE; ++j) {
                    const float4 cam = scam[j];
                    const f2v sc = {cam.z, cam.w}, cns = {cam.w, -cam.z}, camxz = {cam.x, cam.y};
                    unsigned char* const out = img + (size_t)(c0 + j) * ((size_t)p.gpe * 12) + (size_t)cg * 12;
                    f2v rt = lrow[vstart < p.H ? vstart : 0];
                    for (int v = vstart; v < p.H; v += p.rows_per_pass) {
                        const int vn = v + p.rows_per_pass;
                        const f2v rtn = lrow[vn < p.H ? vn : v];                            // prefetch the next row's table entry
                        const unsigned pal_a = lds0 + (unsigned)p.off_pal + ((unsigned)v << 4);
                        const f2v lz2 = {rt.x, rt.x}, kk2 = {rt.y, rt.y};
                        const f2v a = __builtin_elementwise_fma(lz2, sc, camxz);           // (ax, az)
                        const f2v d = kk2 * cns;                                           // (dx, dz) = (k*c, -(k*s))
                        auto shade = [&](f2v uf) -> uint32_t {
                            const f2v g = __builtin_elementwise_fma(uf, d, a);             // (gx, gz)
                            const unsigned ix = min(cvt_u32_sat(g.x), gwm1);
                            const unsigned iz = min(cvt_u32_sat(g.y), ghm1);
                            const unsigned xoff = (ix >> 2) & ~3u;                          // byte offset of the map word in its row
                            unsigned waddr, paddr;
                            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(waddr) : "v"(iz), "s"(pitch), "v"(xoff));
                            const uint32_t w = *(lds_u32p)(uintptr_t)waddr;                 // map lives at LDS offset 0 (checked above)
                            const uint32_t cls = __builtin_amdgcn_ubfe(w, ix << 1, 2);      // offset uses bits [4:0] = 2*(ix&15)
                            asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(paddr) : "v"(cls), "v"(pal_a));
                            return *(lds_u32p)(uintptr_t)paddr;
                        };
#if TRS_ABLATE == 2   /* diagnostic build: stores only */
                        const uint32_t c0p = (uint32_t)v, c1p = c0p + 1, c2p = c0p + 2, c3p = c0p + 3; (void)shade;
#else
                        const uint32_t c0p = shade(ufa), c1p = shade(ufb), c2p = shade(ufc), c3p = shade(ufd);
#endif
                        // 4 x 0x00BBGGRR -> 12 bytes R,G,B,R,G,B,...  (v_perm_b32: selector bytes 0-3 = 2nd operand, 4-7 = 1st)
                        const uint32_t w0 = __builtin_amdgcn_perm(c1p, c0p, 0x04020100u);
                        const uint32_t w1 = __builtin_amdgcn_perm(c2p, c1p, 0x05040201u);
                        const uint32_t w2 = __builtin_amdgcn_perm(c3p, c2p, 0x06050402u);
                        uint32_t* o = reinterpret_cast<uint32_t*>(out + (size_t)v * row_bytes);
#if TRS_ABLATE == 1   /* diagnostic build: compute, no stores */
                        asm volatile("" :: "v"(w0), "v"(w1), "v"(w2)); (void)o;
#else
                        o[0] = w0; o[1] = w1; o[2] = w2;
#endif
                        rt = rtn;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// Batched LocationTracker.__find_closest (components/track_data_process.py:89-101): one wave per query,
// track staged in LDS once per workgroup, queries grid-strided.
__global__ __launch_bounds__(kLocBlock) void trs_locate_kernel(const unsigned char* blob, int pts_bytes, int off_py, int off_pz, int np,
                                                               const double* q, int nq, int32_t* out)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        const uint4* src = reinterpret_cast<const uint4*>(blob);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (int i = tid; i < (pts_bytes >> 4); i += kLocBlock) dst[i] = src[i];
    }
    __syncthreads();
    const double* lpx = reinterpret_cast<const double*>(smem);
    const double* lpy = reinterpret_cast<const double*>(smem + off_py);
    const double* lpz = reinterpret_cast<const double*>(smem + off_pz);
    constexpr int kW = kLocBlock / 64;
    for (int qi = blockIdx.x * kW + wave; qi < nq; qi += gridDim.x * kW) {
        const double qx = q[3 * qi], qy = q[3 * qi + 1], qz = q[3 * qi + 2];
        double best = TRS_LOST_L1;
        int bi = 0;
        for (int i = lane; i < np; i += 64) {
            const double d = (fabs(qx - lpx[i]) + fabs(qy - lpy[i])) + fabs(qz - lpz[i]);
            if (d < best) { best = d; bi = i; }
        }
        wave_argmin(best, bi);
        if (lane == 63) out[qi] = bi;      // wave_argmin leaves the result in lane 63
    }
}

// ---------------------------------------------------------------------------------------------
// host side

thread_local std::string g_err;

int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIPCHK(call)                                                                              \
    do {                                                                                          \
        hipError_t _e = (call);                                                                   \
        if (_e != hipSuccess) return fail(TRS_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(_e)); \
    } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

struct trs_env {
    trs_config cfg{};
    int device = 0, n = 0, H = 0, W = 0, cu_count = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[8] = {};
    // device memory
    unsigned char* slab = nullptr;       // state + controls
    uint8_t* img[2] = {nullptr, nullptr};
    unsigned char* blob = nullptr;
    float* tangent = nullptr;
    float* start_yaw = nullptr;
    unsigned long long* stats = nullptr;
    double* loc_q = nullptr; int32_t* loc_out = nullptr; int loc_cap = 0;
    KParams kp{};
    trsim::TrackTables tab;
    bool track_loaded = false;
    int lds_bytes = 0, pts_bytes = 0;
    uint64_t step_count = 0;
    float *ctl_steer = nullptr, *ctl_thr = nullptr, *ctl_brk = nullptr;
    uint8_t* ctl_reset = nullptr;
    size_t img_bytes = 0;
};

namespace {

int launch_steps(trs_env* e, const float* st, const float* th, const float* br, const uint8_t* rs, int n_steps, int synth)
{
    KParams p = e->kp;
    p.ctl_steer = st; p.ctl_thr = th; p.ctl_brk = br; p.ctl_reset = rs;
    p.synth = synth; p.n_steps = n_steps;
    p.step0 = (uint32_t)e->step_count;
    p.img_parity = (int)(e->step_count & 1);
    const int grid = (e->n + p.envs_per_wg - 1) / p.envs_per_wg;
    hipLaunchKernelGGL(trs_step_kernel, dim3(grid), dim3(kBlock), e->lds_bytes, e->stream, p);
    HIPCHK(hipGetLastError());
    e->step_count += (uint64_t)n_steps;
    return TRS_OK;
}

}  // namespace

TRS_EXPORT void trs_default_config(trs_config* c)
{
    if (!c) return;
    std::memset(c, 0, sizeof *c);
    c->struct_size = (uint32_t)sizeof *c;
    c->n_envs = 1; c->img_h = 120; c->img_w = 160; c->render = 1;
    c->seed = TRS_SYNTH_SEED;
    c->dt = TRS_DEF_DT; c->max_steer = TRS_DEF_MAX_STEER; c->inv_wheelbase = TRS_DEF_INV_WHEELBASE;
    c->accel_max = TRS_DEF_ACCEL_MAX; c->drag_lin = TRS_DEF_DRAG_LIN; c->roll_res = TRS_DEF_ROLL_RES;
    c->brake_max = TRS_DEF_BRAKE_MAX; c->v_max = TRS_DEF_V_MAX; c->v_rev_max = TRS_DEF_V_REV_MAX;
    c->offtrack_cte = TRS_DEF_OFFTRACK_CTE; c->offtrack_penalty = TRS_DEF_OFFTRACK_PENALTY; c->cam_fwd = TRS_DEF_CAM_FWD;
    c->road_half = TRS_DEF_ROAD_HALF; c->edge_half = TRS_DEF_EDGE_HALF; c->centre_half = TRS_DEF_CENTRE_HALF;
    c->dash_period = TRS_DEF_DASH_PERIOD; c->dash_on = TRS_DEF_DASH_ON; c->map_margin = TRS_DEF_MAP_MARGIN;
    c->fov_v_deg = TRS_DEF_FOV_V_DEG; c->cam_h = TRS_DEF_CAM_H; c->cam_pitch_deg = TRS_DEF_CAM_PITCH_DEG; c->z_far = TRS_DEF_Z_FAR;
}

TRS_EXPORT int trs_device_count(int* out)
{
    if (!out) return fail(TRS_ERR_ARG, "null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *out = n;
    return TRS_OK;
}

TRS_EXPORT int trs_create(const trs_config* cfg, int device, trs_env** out)
{
    if (!cfg || !out) return fail(TRS_ERR_ARG, "null argument");
    if (cfg->struct_size != sizeof(trs_config)) return fail(TRS_ERR_ARG, "trs_config.struct_size mismatch");
    if (cfg->n_envs < 1 || cfg->img_h < 2 || cfg->img_w < 4 || (cfg->img_w & 3) || cfg->env_id_base < 0)
        return fail(TRS_ERR_ARG, "bad n_envs / image size (img_w must be a multiple of 4)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(TRS_ERR_DEVICE, "no HIP device visible (libtrsim has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(TRS_ERR_ARG, "device index out of range");
    HIPCHK(hipSetDevice(device));
    trs_env* e = new (std::nothrow) trs_env();
    if (!e) return fail(TRS_ERR_NOMEM, "out of memory");
    e->cfg = *cfg; e->device = device; e->n = cfg->n_envs; e->H = cfg->img_h; e->W = cfg->img_w;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    e->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    for (auto& ev : e->ev) HIPCHK(hipEventCreate(&ev));

    // one slab for all per-env arrays: 12 float + 2 int32 arrays, 3 float + 1 byte control arrays, 2+1 byte arrays
    const size_t n = (size_t)e->n, fa = align_up(n * 4, 256), ba = align_up(n, 256);
    const size_t slab_bytes = fa * (10 + 2 + 3) + ba * 3;
    HIPCHK(hipMalloc((void**)&e->slab, slab_bytes));
    HIPCHK(hipMemsetAsync(e->slab, 0, slab_bytes, e->stream));
    unsigned char* c = e->slab;
    auto takef = [&](float*& ptr) { ptr = reinterpret_cast<float*>(c); c += fa; };
    KParams& k = e->kp;
    takef(k.x); takef(k.y); takef(k.z); takef(k.yaw); takef(k.v); takef(k.speed); takef(k.cte);
    takef(k.ep_return); takef(k.last_return); takef(k.steer_filt);
    k.seg_idx = reinterpret_cast<int32_t*>(c); c += fa;
    k.ep_len = reinterpret_cast<int32_t*>(c); c += fa;
    takef(e->ctl_steer); takef(e->ctl_thr); takef(e->ctl_brk);
    k.done = c; c += ba; k.pending = c; c += ba; e->ctl_reset = c; c += ba;
    HIPCHK(hipMalloc((void**)&e->stats, 4 * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(e->stats, 0, 4 * sizeof(unsigned long long), e->stream));
    k.stats = e->stats;
    if (cfg->render) {
        e->img_bytes = n * (size_t)e->H * e->W * 3;
        for (int b = 0; b < 2; ++b) {
            HIPCHK(hipMalloc((void**)&e->img[b], e->img_bytes));
            HIPCHK(hipMemsetAsync(e->img[b], 0, e->img_bytes, e->stream));
            k.img[b] = e->img[b];
        }
    }
    k.n_envs = e->n; k.env_id_base = cfg->env_id_base;
    k.envs_per_wg = (e->n + e->cu_count - 1) / e->cu_count;
    k.H = e->H; k.W = e->W; k.gpr = e->W / 4; k.gpe = k.gpr * e->H;
    k.dt = cfg->dt; k.max_steer = cfg->max_steer; k.inv_wheelbase = cfg->inv_wheelbase; k.accel_max = cfg->accel_max;
    k.drag_lin = cfg->drag_lin; k.roll_res = cfg->roll_res; k.brake_max = cfg->brake_max; k.v_max = cfg->v_max;
    k.v_rev_max = cfg->v_rev_max; k.offtrack_cte = cfg->offtrack_cte; k.offtrack_penalty = cfg->offtrack_penalty;
    k.cam_fwd = cfg->cam_fwd; k.auto_reset = cfg->auto_reset; k.render = cfg->render; k.seed = cfg->seed;
    if (k.gpr > kBlock) { trs_destroy(e); return fail(TRS_ERR_LIMIT, "img_w too large: more 4-pixel groups per row than threads per workgroup"); }
    k.rows_per_pass = kBlock / k.gpr;      // threads beyond rows_per_pass * gpr idle in the raster phase (none at W = 160: 24 x 40 = 960)
    k.row_magic = 0;
    HIPCHK(hipStreamSynchronize(e->stream));
    *out = e;
    return TRS_OK;
}

TRS_EXPORT int trs_destroy(trs_env* e)
{
    if (!e) return TRS_OK;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    (void)hipFree(e->slab); (void)hipFree(e->img[0]); (void)hipFree(e->img[1]); (void)hipFree(e->blob); (void)hipFree(e->tangent); (void)hipFree(e->start_yaw);
    (void)hipFree(e->stats); (void)hipFree(e->loc_q); (void)hipFree(e->loc_out);
    for (auto& ev : e->ev) if (ev) (void)hipEventDestroy(ev);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return TRS_OK;
}

TRS_EXPORT int trs_load_track(trs_env* e, const double* h_xyz, int n_points)
{
    if (!e) return fail(TRS_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(e->device));
    std::string err;
    int rc = trsim::build_tables(e->cfg, h_xyz, n_points, e->tab, err);
    if (rc) return fail(rc, err);
    const trsim::TrackTables& T = e->tab;
    KParams& k = e->kp;
    // LDS image layout: [map (pitched rows) @0][px][py][pz][tangent][rowtab][palette] + scratch
    const size_t pts = align_up((size_t)n_points * 8, 16);
    const int pitch_words = T.info.map_words | 1;          // odd pitch: rows of the map start on different LDS banks
    k.map_pitch_b = pitch_words * 4;
    if (k.map_pitch_b >= (1 << 24) || T.info.map_h >= (1 << 24)) return fail(TRS_ERR_LIMIT, "map exceeds the 24-bit multiply of the rasteriser");
    const size_t map_bytes = (size_t)k.map_pitch_b * T.info.map_h;
    size_t off = 0;
    k.off_map = 0; off += align_up(map_bytes, 16);
    k.off_px = (int)off; off += pts;
    k.off_py = (int)off; off += pts;
    k.off_pz = (int)off; off += pts;
    e->pts_bytes = (int)(3 * pts);
    k.off_tan = (int)off;
    {   // tangents ride in LDS when everything still fits in the CU's 160 KiB (generated track: yes; mountain track: no)
        const size_t scratch_est = (size_t)kEMax * 16 + 3 * kEMax * 8 + (size_t)kEMax * kWaves * 12 + 64;
        const size_t with_tan = off + align_up((size_t)n_points * 8, 16) + align_up((size_t)e->H * 8, 16) + (size_t)e->H * 16 + scratch_est;
        k.tan_in_lds = with_tan <= 160 * 1024 ? 1 : 0;
        if (k.tan_in_lds) off += align_up((size_t)n_points * 8, 16);
    }
    k.off_rowtab = (int)off; off += align_up((size_t)e->H * 8, 16);
    k.off_pal = (int)off; off += (size_t)e->H * 16;
    k.blob_bytes = (int)off;
    k.off_scratch = (int)off;
    k.stage_bytes = k.blob_bytes;
    if ((size_t)(k.off_scratch - k.off_px) > (size_t)kHotRegs * kStagers * 16 || (size_t)k.off_px > (size_t)kMapRegs * kStagers * 16)
        return fail(TRS_ERR_LIMIT, "tables exceed the register-staging capacity of the step kernel");
    const size_t scratch = (size_t)kEMax * 16 + 3 * kEMax * 8 + (size_t)kEMax * kWaves * 8 + (size_t)kEMax * kWaves * 4;
    e->lds_bytes = (int)align_up(off + scratch, 16);
    if (e->lds_bytes > 160 * 1024) return fail(TRS_ERR_LIMIT, "tables exceed the 160 KiB LDS of a CU");
    std::vector<unsigned char> h(off, 0);
    for (int r = 0; r < T.info.map_h; ++r)
        std::memcpy(h.data() + (size_t)r * k.map_pitch_b, T.map.data() + (size_t)r * T.info.map_words, (size_t)T.info.map_words * 4);
    std::memcpy(h.data() + k.off_px, T.px.data(), (size_t)n_points * 8);
    std::memcpy(h.data() + k.off_py, T.py.data(), (size_t)n_points * 8);
    std::memcpy(h.data() + k.off_pz, T.pz.data(), (size_t)n_points * 8);
    if (k.tan_in_lds) std::memcpy(h.data() + k.off_tan, T.tangent.data(), (size_t)n_points * 8);
    std::memcpy(h.data() + k.off_rowtab, T.rowtab.data(), (size_t)e->H * 8);
    std::memcpy(h.data() + k.off_pal, T.palette.data(), (size_t)e->H * 16);
    HIPCHK(hipStreamSynchronize(e->stream));
    (void)hipFree(e->blob); (void)hipFree(e->tangent); (void)hipFree(e->start_yaw);
    e->blob = nullptr; e->tangent = nullptr; e->start_yaw = nullptr;
    HIPCHK(hipMalloc((void**)&e->blob, off));
    HIPCHK(hipMalloc((void**)&e->tangent, (size_t)n_points * 8));
    HIPCHK(hipMalloc((void**)&e->start_yaw, (size_t)n_points * 4));
    HIPCHK(hipMemcpy(e->blob, h.data(), off, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->tangent, T.tangent.data(), (size_t)n_points * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->start_yaw, T.start_yaw.data(), (size_t)n_points * 4, hipMemcpyHostToDevice));
    k.blob = e->blob; k.start_yaw = e->start_yaw; k.tangent_g = e->tangent;
    k.np = n_points; k.map_w = T.info.map_w; k.map_h = T.info.map_h; k.map_words = T.info.map_words;
    k.map_x0f = T.map_x0f; k.map_z0f = T.map_z0f; k.inv_cellf = T.inv_cellf;
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, e->lds_bytes));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_locate_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, e->pts_bytes));
    // start poses (host mirror of the reset branch so that telemetry is meaningful before the first step)
    const size_t n = (size_t)e->n;
    std::vector<float> sx(n), sy(n), sz(n), syaw(n);
    std::vector<int32_t> sidx(n);
    for (size_t i = 0; i < n; ++i) {
        const int gid = e->cfg.env_id_base + (int)i;
        const int si = (int)(((long long)TRS_START_STRIDE * gid) % n_points);
        sx[i] = (float)T.px[si]; sy[i] = (float)T.py[si]; sz[i] = (float)T.pz[si]; syaw[i] = T.start_yaw[si]; sidx[i] = si;
    }
    HIPCHK(hipMemcpy(k.x, sx.data(), n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(k.y, sy.data(), n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(k.z, sz.data(), n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(k.yaw, syaw.data(), n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(k.seg_idx, sidx.data(), n * 4, hipMemcpyHostToDevice));
    for (float* a : {k.v, k.speed, k.cte, k.ep_return, k.last_return, k.steer_filt}) HIPCHK(hipMemset(a, 0, n * 4));
    HIPCHK(hipMemset(k.ep_len, 0, n * 4));
    HIPCHK(hipMemset(k.done, 0, n));
    HIPCHK(hipMemset(k.pending, 1, n));
    HIPCHK(hipMemset(e->stats, 0, 32));
    e->step_count = 0;
    e->track_loaded = true;
    return TRS_OK;
}

TRS_EXPORT int trs_reset(trs_env* e, const uint8_t* h_mask)
{
    if (!e || !e->track_loaded) return fail(TRS_ERR_STATE, "no track loaded");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (!h_mask) { HIPCHK(hipMemset(e->kp.pending, 1, (size_t)e->n)); return TRS_OK; }
    std::vector<uint8_t> cur((size_t)e->n);
    HIPCHK(hipMemcpy(cur.data(), e->kp.pending, cur.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < e->n; ++i) if (h_mask[i]) cur[i] = 1;
    HIPCHK(hipMemcpy(e->kp.pending, cur.data(), cur.size(), hipMemcpyHostToDevice));
    return TRS_OK;
}

TRS_EXPORT int trs_step(trs_env* e, const float* d_st, const float* d_th, const float* d_br, const uint8_t* d_rs, int n_steps)
{
    if (!e || !e->track_loaded) return fail(TRS_ERR_STATE, "no track loaded");
    if (n_steps < 1) return fail(TRS_ERR_ARG, "n_steps < 1");
    if (!d_st || !d_th) return fail(TRS_ERR_ARG, "null controls");
    HIPCHK(hipSetDevice(e->device));
    return launch_steps(e, d_st, d_th, d_br, d_rs, n_steps, 0);
}

TRS_EXPORT int trs_step_host(trs_env* e, const float* h_st, const float* h_th, const float* h_br, const uint8_t* h_rs, int n_steps)
{
    if (!e || !e->track_loaded) return fail(TRS_ERR_STATE, "no track loaded");
    if (n_steps < 1) return fail(TRS_ERR_ARG, "n_steps < 1");
    if (!h_st || !h_th) return fail(TRS_ERR_ARG, "null controls");
    HIPCHK(hipSetDevice(e->device));
    const size_t n = (size_t)e->n;
    HIPCHK(hipMemcpyAsync(e->ctl_steer, h_st, n * 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->ctl_thr, h_th, n * 4, hipMemcpyHostToDevice, e->stream));
    if (h_br) HIPCHK(hipMemcpyAsync(e->ctl_brk, h_br, n * 4, hipMemcpyHostToDevice, e->stream));
    if (h_rs) HIPCHK(hipMemcpyAsync(e->ctl_reset, h_rs, n, hipMemcpyHostToDevice, e->stream));
    return launch_steps(e, e->ctl_steer, e->ctl_thr, h_br ? e->ctl_brk : nullptr, h_rs ? e->ctl_reset : nullptr, n_steps, 0);
}

TRS_EXPORT int trs_step_synthetic(trs_env* e, int n_steps, int steps_per_launch)
{
    if (!e || !e->track_loaded) return fail(TRS_ERR_STATE, "no track loaded");
    if (n_steps < 1 || steps_per_launch < 1) return fail(TRS_ERR_ARG, "n_steps / steps_per_launch < 1");
    HIPCHK(hipSetDevice(e->device));
    for (int done = 0; done < n_steps;) {
        const int now = std::min(steps_per_launch, n_steps - done);
        int rc = launch_steps(e, nullptr, nullptr, nullptr, nullptr, now, 1);
        if (rc) return rc;
        done += now;
    }
    return TRS_OK;
}

TRS_EXPORT int trs_get_state(trs_env* e, trs_state_view* o)
{
    if (!e || !o) return fail(TRS_ERR_ARG, "null argument");
    const KParams& k = e->kp;
    o->n_envs = e->n; o->img_h = e->H; o->img_w = e->W; o->n_points = k.np;
    o->img = e->cfg.render ? e->img[(e->step_count + 1) & 1] : nullptr;   // buffer written by the last step
    o->pos_x = k.x; o->pos_y = k.y; o->pos_z = k.z; o->speed = k.speed; o->cte = k.cte; o->yaw = k.yaw; o->vel = k.v;
    o->seg_idx = k.seg_idx; o->ep_return = k.ep_return; o->last_return = k.last_return; o->ep_len = k.ep_len; o->done = k.done;
    o->step_count = e->step_count;
    return TRS_OK;
}

TRS_EXPORT int trs_copy_to_host(trs_env* e, int which, void* dst, size_t bytes)
{
    if (!e || !dst) return fail(TRS_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(e->device));
    const KParams& k = e->kp;
    const void* src = nullptr; size_t need = 0; const size_t n = (size_t)e->n;
    bool host_src = false;
    switch (which) {
    case TRS_F_IMG: src = e->cfg.render ? e->img[(e->step_count + 1) & 1] : nullptr; need = e->img_bytes; break;
    case TRS_F_POS_X: src = k.x; need = n * 4; break;
    case TRS_F_POS_Y: src = k.y; need = n * 4; break;
    case TRS_F_POS_Z: src = k.z; need = n * 4; break;
    case TRS_F_SPEED: src = k.speed; need = n * 4; break;
    case TRS_F_CTE: src = k.cte; need = n * 4; break;
    case TRS_F_YAW: src = k.yaw; need = n * 4; break;
    case TRS_F_VEL: src = k.v; need = n * 4; break;
    case TRS_F_SEG_IDX: src = k.seg_idx; need = n * 4; break;
    case TRS_F_EP_RETURN: src = k.ep_return; need = n * 4; break;
    case TRS_F_LAST_RETURN: src = k.last_return; need = n * 4; break;
    case TRS_F_EP_LEN: src = k.ep_len; need = n * 4; break;
    case TRS_F_DONE: src = k.done; need = n; break;
    case TRS_F_STEER_FILT: src = k.steer_filt; need = n * 4; break;
    case TRS_F_MAP: if (e->track_loaded) { src = e->tab.map.data(); need = (size_t)k.map_words * k.map_h * 4; host_src = true; } break;
    case TRS_F_ROWTAB: if (e->track_loaded) { src = e->blob + k.off_rowtab; need = (size_t)e->H * 8; } break;
    case TRS_F_PALETTE: if (e->track_loaded) { src = e->blob + k.off_pal; need = (size_t)e->H * 16; } break;
    case TRS_F_TANGENT: if (e->track_loaded) { src = e->tangent; need = (size_t)k.np * 8; } break;
    default: return fail(TRS_ERR_ARG, "unknown field");
    }
    if (!src) return fail(TRS_ERR_STATE, "field not available");
    if (bytes != need) return fail(TRS_ERR_ARG, "byte count mismatch");
    if (host_src) { std::memcpy(dst, src, need); return TRS_OK; }   // the unpitched map lives on the host; its LDS image is pitched
    HIPCHK(hipMemcpyAsync(dst, src, need, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return TRS_OK;
}

TRS_EXPORT int trs_set_pose(trs_env* e, const float* x, const float* y, const float* z, const float* yaw, const float* v)
{
    if (!e || !e->track_loaded) return fail(TRS_ERR_STATE, "no track loaded");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    const size_t n = (size_t)e->n;
    const KParams& k = e->kp;
    if (x) HIPCHK(hipMemcpy(k.x, x, n * 4, hipMemcpyHostToDevice));
    if (y) HIPCHK(hipMemcpy(k.y, y, n * 4, hipMemcpyHostToDevice));
    if (z) HIPCHK(hipMemcpy(k.z, z, n * 4, hipMemcpyHostToDevice));
    if (yaw) HIPCHK(hipMemcpy(k.yaw, yaw, n * 4, hipMemcpyHostToDevice));
    if (v) HIPCHK(hipMemcpy(k.v, v, n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(k.pending, 0, n));
    HIPCHK(hipMemset(k.done, 0, n));
    return TRS_OK;
}

TRS_EXPORT int trs_locate(trs_env* e, const double* h_xyz, int nq, int32_t* h_idx)
{
    if (!e || !e->track_loaded) return fail(TRS_ERR_STATE, "no track loaded");
    if (nq < 0 || (nq && (!h_xyz || !h_idx))) return fail(TRS_ERR_ARG, "bad query");
    if (nq == 0) return TRS_OK;
    HIPCHK(hipSetDevice(e->device));
    if (nq > e->loc_cap) {
        HIPCHK(hipStreamSynchronize(e->stream));
        (void)hipFree(e->loc_q); (void)hipFree(e->loc_out); e->loc_q = nullptr; e->loc_out = nullptr; e->loc_cap = 0;
        HIPCHK(hipMalloc((void**)&e->loc_q, (size_t)nq * 24));
        HIPCHK(hipMalloc((void**)&e->loc_out, (size_t)nq * 4));
        e->loc_cap = nq;
    }
    HIPCHK(hipMemcpyAsync(e->loc_q, h_xyz, (size_t)nq * 24, hipMemcpyHostToDevice, e->stream));
    constexpr int kW = kLocBlock / 64;
    int grid = (nq + kW - 1) / kW;
    grid = std::min(grid, e->cu_count * 2);
    hipLaunchKernelGGL(trs_locate_kernel, dim3(grid), dim3(kLocBlock), e->pts_bytes, e->stream,
                       (const unsigned char*)e->blob + e->kp.off_px, e->pts_bytes, e->kp.off_py - e->kp.off_px, e->kp.off_pz - e->kp.off_px, e->kp.np,
                       (const double*)e->loc_q, nq, e->loc_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h_idx, e->loc_out, (size_t)nq * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return TRS_OK;
}

TRS_EXPORT int trs_map_info_get(trs_env* e, trs_map_info* o)
{
    if (!e || !o || !e->track_loaded) return fail(TRS_ERR_STATE, "no track loaded");
    *o = e->tab.info;
    o->lds_bytes = e->lds_bytes;
    return TRS_OK;
}

TRS_EXPORT int trs_sync(trs_env* e)
{
    if (!e) return fail(TRS_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    return TRS_OK;
}

TRS_EXPORT int trs_event_record(trs_env* e, int slot)
{
    if (!e || slot < 0 || slot >= 8) return fail(TRS_ERR_ARG, "bad event slot");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipEventRecord(e->ev[slot], e->stream));
    return TRS_OK;
}

TRS_EXPORT int trs_event_elapsed_ms(trs_env* e, int a, int b, float* ms)
{
    if (!e || !ms || a < 0 || a >= 8 || b < 0 || b >= 8) return fail(TRS_ERR_ARG, "bad event slot");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipEventSynchronize(e->ev[b]));
    HIPCHK(hipEventElapsedTime(ms, e->ev[a], e->ev[b]));
    return TRS_OK;
}

TRS_EXPORT const char* trs_last_error(void) { return g_err.c_str(); }
