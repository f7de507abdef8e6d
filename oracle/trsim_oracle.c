/* trsim_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (triton-racer-sim_amd/) never does and fails loudly without its
 * HIP extension.
 *
 * What it restates, and how it is pinned:
 *   - trso_locate / the nearest-point search: a plain-C restatement of the reference's
 *     LocationTracker.__find_closest / __distance
 *     (/root/reference/TritonRacerSim/components/track_data_process.py:89-104).
 *     PINNED by tests/golden/locate_*.json, captured from the reference itself
 *     (tests/golden/gen_golden.py).
 *   - bicycle-model step, cross-track error, class map, camera: there is NO reference
 *     implementation (the reference talks to a closed Unity binary,
 *     components/gyminterface.py:47-104).  These follow the text of
 *     include/trsim_spec.h.  PARITY UNPINNED against the reference for these parts.
 *
 * Scalar, one env at a time, written for clarity; `#pragma omp` over envs only so the
 * CPU baseline can use all host cores.  Exports the trsim.h ABI with prefix trso_.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/trsim.h"
#include "../include/trsim_spec.h"

#define EXPORT __attribute__((visibility("default")))

static __thread char g_err[256];
static int fail(int code, const char* msg) { snprintf(g_err, sizeof g_err, "%s", msg); return code; }

struct mux_car;
struct trs_env {
    trs_config cfg;
    int n, H, W;
    int threads;
    /* track */
    int np;
    double *px, *py, *pz;      /* raw points, binary64 */
    float *tang;               /* [np][2] */
    float *start_yaw;          /* [np] */
    /* map + tables */
    trs_map_info mi;
    uint32_t* map;
    float* rowtab;             /* [H][2] */
    uint32_t* pal;             /* [H][4] */
    float* rowdepth;           /* [H] */
    float* depth;              /* [n][H][W] when cfg.depth */
    /* tracks with elevation (include/trsim_spec.h): per-point view-pitch offsets and what the per-env row tables need */
    int hills;
    float* dpitch;             /* [np] */
    uint32_t* sky;             /* [H] sky colour of every row */
    uint32_t far_rgb;
    float inv_f, hh, pitch_f, cam_h_f, z_far_f, inv_zfar_f, fog_f;
    float map_x0f, map_z0f, inv_cellf;
    /* state */
    float *x, *y, *z, *yaw, *v, *speed, *cte, *ep_return, *last_return, *steer_filt;
    int32_t *seg_idx, *ep_len;
    uint8_t *done, *pending, *was_reset;
    uint8_t* img;
    uint8_t* pre;              /* processed-image buffer (trso_preprocess with dst == NULL) */
    trs_pre_config frame_filter; int has_frame_filter;   /* trso_set_frame_filter */
    struct mux_car* mux;       /* ControlMultiplexer state per car (trso_control_mux) */
    int mux_tick;
    uint64_t step_count;
    uint64_t stats[64];        /* [0] off-track events, [1] resets */
    int comm_ready;            /* trso_comm_init was called (one rank) */
    void* scratch[32]; size_t scratch_bytes[32];   /* trso_scratch */
};

/* ------------------------------------------------------------------ spec pieces */

static void spec_sincos(float a, float* so, float* co)
{
    float q = rintf(a * TRS_TWO_OVER_PI);
    float r = fmaf(q, -TRS_PIO2_HI, a);
    r = fmaf(q, -TRS_PIO2_LO, r);
    float z = r * r;
    float ps = fmaf(fmaf(TRS_S0, z, TRS_S1), z, TRS_S2);
    float s = fmaf(r * z, ps, r);
    float pc = fmaf(fmaf(TRS_C0, z, TRS_C1), z, TRS_C2);
    float c = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
    switch (((int)q) & 3) {
    case 0: *so = s;  *co = c;  break;
    case 1: *so = c;  *co = -s; break;
    case 2: *so = -s; *co = -c; break;
    default: *so = -c; *co = s; break;
    }
}

static float clampf(float a, float lo, float hi) { return a < lo ? lo : (a > hi ? hi : a); }

/* track_data_process.py:89-104: sequential scan, best = 100, strict '<'. */
static int l1_nearest(const struct trs_env* e, double qx, double qy, double qz, double* best_out)
{
    int sel = 0;
    double best = TRS_LOST_L1;
    for (int i = 0; i < e->np; ++i) {
        double d = fabs(qx - e->px[i]) + fabs(qy - e->py[i]) + fabs(qz - e->pz[i]);
        if (d < best) { sel = i; best = d; }
    }
    if (best_out) *best_out = best;
    return sel;
}

static void synth_controls(uint64_t seed, uint32_t gid, uint32_t step, float* sf, float* steer, float* thr)
{
    uint64_t z = seed + (((uint64_t)gid << 32) | (uint64_t)step) * 0x9E3779B97F4A7C15ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    float us = (float)(uint32_t)(z >> 40) * 5.9604644775390625e-08f;
    float ut = (float)(uint32_t)((z >> 16) & 0xFFFFFFu) * 5.9604644775390625e-08f;
    float raw = us * 2.0f - 1.0f;
    float nsf = *sf + TRS_SYNTH_ALPHA * (raw - *sf);
    *sf = nsf;
    *steer = nsf;
    *thr = TRS_SYNTH_THR_LO + TRS_SYNTH_THR_SPAN * ut;
}

/* one env, one step; returns heading (s, c) for the camera */
static void step_env(struct trs_env* e, int i, float steer, float thr, float brk, int reset_in, float* hs, float* hc)
{
    const trs_config* k = &e->cfg;
    int do_reset = e->pending[i] || reset_in || (k->auto_reset && e->done[i]);
    float x1, z1, yaw1, v2, s, c;
    if (do_reset) {
        int gid = k->env_id_base + i;
        int si = (int)(((int64_t)TRS_START_STRIDE * gid) % e->np);
        e->last_return[i] = e->ep_return[i];
        e->ep_return[i] = 0.0f;
        e->ep_len[i] = 0;
        e->steer_filt[i] = 0.0f;
        e->pending[i] = 0;
        e->was_reset[i] = 1;
        x1 = (float)e->px[si]; e->y[i] = (float)e->py[si]; z1 = (float)e->pz[si];
        yaw1 = e->start_yaw[si]; v2 = 0.0f;
        spec_sincos(yaw1, &s, &c);
    } else {
        e->was_reset[i] = 0;
        steer = clampf(steer, -1.0f, 1.0f);
        thr = clampf(thr, -1.0f, 1.0f);
        brk = clampf(brk, 0.0f, 1.0f);
        float sd, cd;
        spec_sincos(steer * k->max_steer, &sd, &cd);
        float tan_d = sd / cd;
        float v = e->v[i];
        float a = thr * k->accel_max - k->drag_lin * v;
        float v1 = v + a * k->dt;
        float dv = (k->roll_res + brk * k->brake_max) * k->dt;
        if (v1 > 0.0f) { v2 = v1 - dv; if (v2 < 0.0f) v2 = 0.0f; }
        else if (v1 < 0.0f) { v2 = v1 + dv; if (v2 > 0.0f) v2 = 0.0f; }
        else v2 = 0.0f;
        v2 = clampf(v2, -k->v_rev_max, k->v_max);
        yaw1 = e->yaw[i] + ((v2 * tan_d) * k->inv_wheelbase) * k->dt;
        if (yaw1 > TRS_PI) yaw1 -= TRS_TWO_PI;
        if (yaw1 < -TRS_PI) yaw1 += TRS_TWO_PI;
        spec_sincos(yaw1, &s, &c);
        x1 = e->x[i] + (v2 * s) * k->dt;
        z1 = e->z[i] + (v2 * c) * k->dt;
    }
    double best;
    int idx = l1_nearest(e, (double)x1, (double)e->y[i], (double)z1, &best);
    float y1 = (float)e->py[idx];
    float cte = (x1 - (float)e->px[idx]) * e->tang[2 * idx + 1] - (z1 - (float)e->pz[idx]) * e->tang[2 * idx];
    int lost = best >= TRS_LOST_L1;
    int done = (fabsf(cte) > k->offtrack_cte) || lost;
    if (do_reset) {
        /* first observation of the new episode: no reward */
    } else {
        int d = idx - e->seg_idx[i];
        int half = e->np / 2;
        if (d >= e->np - half) d -= e->np;       /* wrap to [-n/2, n/2) */
        if (d < -half) d += e->np;
        float reward = (float)d - (done ? k->offtrack_penalty : 0.0f);
        e->ep_return[i] = e->ep_return[i] + reward;
        e->ep_len[i] += 1;
    }
    e->x[i] = x1; e->y[i] = y1; e->z[i] = z1; e->yaw[i] = yaw1; e->v[i] = v2;
    e->speed[i] = fabsf(v2); e->cte[i] = cte; e->seg_idx[i] = idx; e->done[i] = (uint8_t)done;
    *hs = s; *hc = c;
}

static void render_env(struct trs_env* e, int i, float s, float c)
{
    const trs_config* k = &e->cfg;
    const int H = e->H, W = e->W, GW = e->mi.map_w, GH = e->mi.map_h, MW = e->mi.map_words;
    float camx = ((e->x[i] + k->cam_fwd * s) - e->map_x0f) * e->inv_cellf;
    float camz = ((e->z[i] + k->cam_fwd * c) - e->map_z0f) * e->inv_cellf;
    uint8_t* out = e->img + (size_t)i * H * W * 3;
    float half_w = (float)(W / 2);
    /* row tables of THIS frame: the host's (flat track), or the env's own from the slope ahead of its track point (hilly track) */
    const float* rowtab = e->rowtab; const uint32_t* palt = e->pal; const float* rowdepth = e->rowdepth;
    float* h_rt = NULL; uint32_t* h_pal = NULL; float* h_dep = NULL;
    if (e->hills) {
        static const int base[4][3] = { TRS_RGB_GRASS, TRS_RGB_ROAD, TRS_RGB_EDGE, TRS_RGB_CENTRE };
        static const int fog[3] = TRS_RGB_FOG;
        h_rt = malloc(sizeof(float) * 2 * H); h_pal = malloc(sizeof(uint32_t) * 4 * H); h_dep = malloc(sizeof(float) * H);
        float P = e->pitch_f + e->dpitch[e->seg_idx[i]], sp, cp;
        spec_sincos(P, &sp, &cp);
        for (int v = 0; v < H; ++v) {
            float yn = (e->hh - ((float)v + 0.5f)) * e->inv_f;
            float dy = yn * cp - sp, dzr = yn * sp + cp;
            float lz = 0.0f, kk = 0.0f, dep = e->z_far_f;
            uint32_t col[4];
            if (dy >= -1e-6f) { for (int cc = 0; cc < 4; ++cc) col[cc] = e->sky[v]; }
            else {
                float t = e->cam_h_f / (-dy);
                float zd = t * dzr;
                if (zd > e->z_far_f) { for (int cc = 0; cc < 4; ++cc) col[cc] = e->far_rgb; }
                else {
                    lz = zd * e->inv_cellf; kk = (t * e->inv_f) * e->inv_cellf; dep = zd;
                    float fw = e->fog_f * (zd * e->inv_zfar_f), om = 1.0f - fw;
                    for (int cc = 0; cc < 4; ++cc) {
                        uint32_t rgb = 0;
                        for (int ch = 0; ch < 3; ++ch) {
                            float a = (float)base[cc][ch] * om, b = (float)fog[ch] * fw;
                            float sum = a + b;
                            rgb |= (uint32_t)(int)(sum + 0.5f) << (8 * ch);
                        }
                        col[cc] = rgb;
                    }
                }
            }
            h_rt[2 * v] = lz; h_rt[2 * v + 1] = kk; h_dep[v] = dep;
            for (int cc = 0; cc < 4; ++cc) h_pal[4 * v + cc] = col[cc];
        }
        rowtab = h_rt; palt = h_pal; rowdepth = h_dep;
    }
    if (e->depth)
        for (int v = 0; v < H; ++v)
            for (int u = 0; u < W; ++u) e->depth[((size_t)i * H + v) * W + u] = rowdepth[v];
    for (int v = 0; v < H; ++v) {
        float lz = rowtab[2 * v], kk = rowtab[2 * v + 1];
        float ax = fmaf(lz, s, camx), az = fmaf(lz, c, camz);
        float dx = kk * c, dz = -(kk * s);
        const uint32_t* pal = palt + 4 * v;
        for (int u = 0; u < W; ++u) {
            float uf = (float)u + 0.5f - half_w;
            float gx = fmaf(uf, dx, ax), gz = fmaf(uf, dz, az);
            int ix = (int)floorf(gx), iz = (int)floorf(gz);
            ix = ix < 0 ? 0 : (ix > GW - 1 ? GW - 1 : ix);
            iz = iz < 0 ? 0 : (iz > GH - 1 ? GH - 1 : iz);
            uint32_t w = e->map[(size_t)iz * MW + (ix >> 4)];
            uint32_t cls = (w >> ((ix & 15) * 2)) & 3u;
            uint32_t rgb = pal[cls];
            out[0] = (uint8_t)(rgb & 255u); out[1] = (uint8_t)((rgb >> 8) & 255u); out[2] = (uint8_t)((rgb >> 16) & 255u);
            out += 3;
        }
    }
    free(h_rt); free(h_pal); free(h_dep);
}

/* ------------------------------------------------------------------ host-side table building */

static int build_track_tables(struct trs_env* e)
{
    const trs_config* k = &e->cfg;
    const int np = e->np;
    /* tangents: next distinct minus previous distinct point in (x, z), closed loop */
    for (int i = 0; i < np; ++i) {
        int j = i, b = i, n;
        for (n = 0; n < np; ++n) { j = (j + 1) % np; if (e->px[j] != e->px[i] || e->pz[j] != e->pz[i]) break; }
        if (n == np) return fail(TRS_ERR_ARG, "track has no two distinct points");
        for (n = 0; n < np; ++n) { b = (b + np - 1) % np; if (e->px[b] != e->px[i] || e->pz[b] != e->pz[i]) break; }
        double tx = e->px[j] - e->px[b], tz = e->pz[j] - e->pz[b];
        double len = sqrt(tx * tx + tz * tz);
        if (len == 0.0) { tx = e->px[j] - e->px[i]; tz = e->pz[j] - e->pz[i]; len = sqrt(tx * tx + tz * tz); }
        tx /= len; tz /= len;
        e->tang[2 * i] = (float)tx; e->tang[2 * i + 1] = (float)tz;
        e->start_yaw[i] = (float)atan2(tx, tz);
    }
    /* de-duplicated closed polyline */
    double* qx = malloc(sizeof(double) * (np + 1)), *qz = malloc(sizeof(double) * (np + 1));
    double* qs = malloc(sizeof(double) * (np + 1));
    int m = 0;
    for (int i = 0; i < np; ++i) {
        if (m && qx[m - 1] == e->px[i] && qz[m - 1] == e->pz[i]) continue;
        qx[m] = e->px[i]; qz[m] = e->pz[i]; ++m;
    }
    if (m > 1 && qx[m - 1] == qx[0] && qz[m - 1] == qz[0]) --m;
    double xmin = qx[0], xmax = qx[0], zmin = qz[0], zmax = qz[0];
    for (int i = 1; i < m; ++i) {
        if (qx[i] < xmin) xmin = qx[i];
        if (qx[i] > xmax) xmax = qx[i];
        if (qz[i] < zmin) zmin = qz[i];
        if (qz[i] > zmax) zmax = qz[i];
    }
    double cell = TRS_MAP_CELL_MIN;
    int GW, GH, MW;
    double x0, z0;
    for (;;) {
        x0 = floor((xmin - k->map_margin) / cell) * cell;
        z0 = floor((zmin - k->map_margin) / cell) * cell;
        GW = (int)ceil((xmax + k->map_margin - x0) / cell);
        GH = (int)ceil((zmax + k->map_margin - z0) / cell);
        MW = (GW + 15) / 16;
        if ((size_t)MW * 4 * GH <= TRS_MAP_LDS_BUDGET) break;
        cell *= 2.0;
        if (cell > 64.0) { free(qx); free(qz); free(qs); return fail(TRS_ERR_LIMIT, "track too large for the map budget"); }
    }
    e->mi.map_w = GW; e->mi.map_h = GH; e->mi.map_words = MW; e->mi.cell = cell; e->mi.x0 = x0; e->mi.z0 = z0;
    e->mi.n_points = np;
    e->map_x0f = (float)x0; e->map_z0f = (float)z0; e->inv_cellf = (float)(1.0 / cell);

    size_t ncell = (size_t)GW * GH;
    double* D2 = malloc(sizeof(double) * ncell), *S = malloc(sizeof(double) * ncell);
    for (size_t c = 0; c < ncell; ++c) { D2[c] = INFINITY; S[c] = 0.0; }
    double reach = k->road_half + k->edge_half + cell;
    double s_acc = 0.0;
    for (int sgm = 0; sgm < m; ++sgm) {
        double ax = qx[sgm], az = qz[sgm], bx = qx[(sgm + 1) % m], bz = qz[(sgm + 1) % m];
        double abx = bx - ax, abz = bz - az;
        double len2 = abx * abx + abz * abz, len = sqrt(len2);
        qs[sgm] = s_acc;
        int ix0 = (int)floor(((ax < bx ? ax : bx) - reach - x0) / cell), ix1 = (int)floor(((ax > bx ? ax : bx) + reach - x0) / cell);
        int iz0 = (int)floor(((az < bz ? az : bz) - reach - z0) / cell), iz1 = (int)floor(((az > bz ? az : bz) + reach - z0) / cell);
        if (ix0 < 0) ix0 = 0;
        if (iz0 < 0) iz0 = 0;
        if (ix1 > GW - 1) ix1 = GW - 1;
        if (iz1 > GH - 1) iz1 = GH - 1;
        for (int iz = iz0; iz <= iz1; ++iz)
            for (int ix = ix0; ix <= ix1; ++ix) {
                double cx = x0 + ((double)ix + 0.5) * cell, cz = z0 + ((double)iz + 0.5) * cell;
                double apx = cx - ax, apz = cz - az;
                double t = (apx * abx + apz * abz) / len2;
                t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
                double ex = cx - (ax + t * abx), ez = cz - (az + t * abz);
                double d2 = ex * ex + ez * ez;
                size_t ci = (size_t)iz * GW + ix;
                if (d2 < D2[ci]) { D2[ci] = d2; S[ci] = s_acc + t * len; }
            }
        s_acc += len;
    }
    e->map = calloc((size_t)MW * GH, sizeof(uint32_t));
    for (int iz = 1; iz < GH - 1; ++iz)
        for (int ix = 1; ix < GW - 1; ++ix) {
            size_t ci = (size_t)iz * GW + ix;
            if (!(D2[ci] < INFINITY)) continue;
            double d = sqrt(D2[ci]);
            uint32_t cls;
            if (d <= k->centre_half && fmod(S[ci], k->dash_period) < k->dash_on) cls = TRS_CLS_CENTRE;
            else if (fabs(d - k->road_half) <= k->edge_half) cls = TRS_CLS_EDGE;
            else if (d < k->road_half) cls = TRS_CLS_ROAD;
            else cls = TRS_CLS_GRASS;
            e->map[(size_t)iz * MW + (ix >> 4)] |= cls << ((ix & 15) * 2);
        }
    free(D2); free(S); free(qx); free(qz); free(qs);

    /* camera rows + palette */
    static const int base[4][3] = { TRS_RGB_GRASS, TRS_RGB_ROAD, TRS_RGB_EDGE, TRS_RGB_CENTRE };
    static const int fog[3] = TRS_RGB_FOG, sky_top[3] = TRS_RGB_SKY_TOP, sky_hor[3] = TRS_RGB_SKY_HOR;
    const int H = e->H;
    const double PI_D = 3.14159265358979323846;
    double f = ((double)H / 2.0) / tan(k->fov_v_deg * PI_D / 180.0 / 2.0);
    double pitch = k->cam_pitch_deg * PI_D / 180.0, cp = cos(pitch), sp = sin(pitch);
    for (int v = 0; v < H; ++v) {
        double yn = ((double)H / 2.0 - ((double)v + 0.5)) / f;
        double dy = yn * cp - sp, dz = yn * sp + cp;
        int rgb[4][3];
        float lz = 0.0f, kk = 0.0f;
        if (dy >= -1e-6) {
            double g = ((double)v + 0.5) / ((double)H / 2.0);
            if (g > 1.0) g = 1.0;
            for (int ch = 0; ch < 3; ++ch) {
                int val = (int)floor((double)sky_top[ch] + ((double)sky_hor[ch] - (double)sky_top[ch]) * g + 0.5);
                for (int c = 0; c < 4; ++c) rgb[c][ch] = val;
            }
        } else {
            double t = k->cam_h / (-dy);
            double fwd = t * dz;
            if (fwd > k->z_far) {
                for (int ch = 0; ch < 3; ++ch) {
                    int val = (int)floor((double)base[0][ch] * (1.0 - TRS_FOG_MAX) + (double)fog[ch] * TRS_FOG_MAX + 0.5);
                    for (int c = 0; c < 4; ++c) rgb[c][ch] = val;
                }
            } else {
                lz = (float)(fwd / cell);
                kk = (float)((t / f) / cell);
                double fw = TRS_FOG_MAX * (fwd / k->z_far);
                for (int c = 0; c < 4; ++c)
                    for (int ch = 0; ch < 3; ++ch)
                        rgb[c][ch] = (int)floor((double)base[c][ch] * (1.0 - fw) + (double)fog[ch] * fw + 0.5);
            }
        }
        e->rowtab[2 * v] = lz; e->rowtab[2 * v + 1] = kk;
        e->rowdepth[v] = (lz != 0.0f || kk != 0.0f) ? (float)((k->cam_h / (-dy)) * dz) : (float)k->z_far;
        for (int c = 0; c < 4; ++c)
            e->pal[4 * v + c] = (uint32_t)rgb[c][0] | ((uint32_t)rgb[c][1] << 8) | ((uint32_t)rgb[c][2] << 16);
    }
    /* tracks with elevation (include/trsim_spec.h): grade over +- L samples, the slope A samples ahead against the slope here */
    {
        double ymin = e->py[0], ymax = e->py[0];
        for (int i = 1; i < np; ++i) { if (e->py[i] < ymin) ymin = e->py[i]; if (e->py[i] > ymax) ymax = e->py[i]; }
        e->hills = (ymax - ymin) > TRS_HILL_MIN_RANGE;
        free(e->dpitch); free(e->sky);
        e->dpitch = calloc((size_t)np, sizeof(float));
        e->sky = calloc((size_t)H, sizeof(uint32_t));
        const int L = TRS_HILL_SPAN, A = TRS_HILL_AHEAD;
        double* hstep = malloc(sizeof(double) * np), *theta = malloc(sizeof(double) * np);
        for (int i = 0; i < np; ++i) {
            int j = (i + 1) % np;
            double ddx = e->px[j] - e->px[i], ddz = e->pz[j] - e->pz[i];
            hstep[i] = sqrt(ddx * ddx + ddz * ddz);
        }
        for (int i = 0; i < np; ++i) {
            double d = 0.0;
            for (int q = -L; q < L; ++q) d += hstep[((i + q) % np + np) % np];
            double g = d > 1e-9 ? (e->py[(i + L) % np] - e->py[((i - L) % np + np) % np]) / d : 0.0;
            theta[i] = atan(g);
        }
        for (int i = 0; i < np; ++i) {
            double dp = theta[(i + A) % np] - theta[i];
            if (dp > TRS_HILL_MAX_DPITCH) dp = TRS_HILL_MAX_DPITCH;
            if (dp < -TRS_HILL_MAX_DPITCH) dp = -TRS_HILL_MAX_DPITCH;
            e->dpitch[i] = e->hills ? (float)dp : 0.0f;
        }
        free(hstep); free(theta);
        for (int v = 0; v < H; ++v) {
            double g = ((double)v + 0.5) / ((double)H / 2.0);
            if (g > 1.0) g = 1.0;
            uint32_t rgbv = 0;
            for (int ch = 0; ch < 3; ++ch)
                rgbv |= (uint32_t)(int)floor((double)sky_top[ch] + ((double)sky_hor[ch] - (double)sky_top[ch]) * g + 0.5) << (8 * ch);
            e->sky[v] = rgbv;
        }
        e->far_rgb = 0;
        for (int ch = 0; ch < 3; ++ch)
            e->far_rgb |= (uint32_t)(int)floor((double)base[0][ch] * (1.0 - TRS_FOG_MAX) + (double)fog[ch] * TRS_FOG_MAX + 0.5) << (8 * ch);
        e->inv_f = (float)(1.0 / f); e->hh = (float)((double)H / 2.0); e->pitch_f = (float)pitch;
        e->cam_h_f = (float)k->cam_h; e->z_far_f = (float)k->z_far; e->inv_zfar_f = (float)(1.0 / k->z_far); e->fog_f = (float)TRS_FOG_MAX;
    }
    return TRS_OK;
}

/* ------------------------------------------------------------------ ABI */

EXPORT void trso_default_config(trs_config* c)
{
    memset(c, 0, sizeof *c);
    c->struct_size = (uint32_t)sizeof *c;
    c->n_envs = 1; c->env_id_base = 0; c->img_h = 120; c->img_w = 160; c->render = 1; c->auto_reset = 0;
    c->seed = TRS_SYNTH_SEED;
    c->dt = TRS_DEF_DT; c->max_steer = TRS_DEF_MAX_STEER; c->inv_wheelbase = TRS_DEF_INV_WHEELBASE;
    c->accel_max = TRS_DEF_ACCEL_MAX; c->drag_lin = TRS_DEF_DRAG_LIN; c->roll_res = TRS_DEF_ROLL_RES;
    c->brake_max = TRS_DEF_BRAKE_MAX; c->v_max = TRS_DEF_V_MAX; c->v_rev_max = TRS_DEF_V_REV_MAX;
    c->offtrack_cte = TRS_DEF_OFFTRACK_CTE; c->offtrack_penalty = TRS_DEF_OFFTRACK_PENALTY; c->cam_fwd = TRS_DEF_CAM_FWD;
    c->road_half = TRS_DEF_ROAD_HALF; c->edge_half = TRS_DEF_EDGE_HALF; c->centre_half = TRS_DEF_CENTRE_HALF;
    c->dash_period = TRS_DEF_DASH_PERIOD; c->dash_on = TRS_DEF_DASH_ON; c->map_margin = TRS_DEF_MAP_MARGIN;
    c->fov_v_deg = TRS_DEF_FOV_V_DEG; c->cam_h = TRS_DEF_CAM_H; c->cam_pitch_deg = TRS_DEF_CAM_PITCH_DEG; c->z_far = TRS_DEF_Z_FAR;
}

EXPORT int trso_create(const trs_config* cfg, int device, trs_env** out)
{
    (void)device;
    if (!cfg || !out) return fail(TRS_ERR_ARG, "null argument");
    if (cfg->struct_size != sizeof(trs_config)) return fail(TRS_ERR_ARG, "trs_config.struct_size mismatch");
    if (cfg->n_envs < 1 || cfg->img_h < 2 || cfg->img_w < 4 || (cfg->img_w & 3) || cfg->env_id_base < 0)
        return fail(TRS_ERR_ARG, "bad n_envs / image size (img_w must be a multiple of 4)");
    struct trs_env* e = calloc(1, sizeof *e);
    if (!e) return fail(TRS_ERR_NOMEM, "out of memory");
    e->cfg = *cfg; e->n = cfg->n_envs; e->H = cfg->img_h; e->W = cfg->img_w; e->threads = 1;
    int n = e->n;
#define A(p, T) p = calloc((size_t)n, sizeof(T))
    A(e->x, float); A(e->y, float); A(e->z, float); A(e->yaw, float); A(e->v, float); A(e->speed, float); A(e->cte, float);
    A(e->ep_return, float); A(e->last_return, float); A(e->steer_filt, float); A(e->seg_idx, int32_t); A(e->ep_len, int32_t);
    A(e->done, uint8_t); A(e->pending, uint8_t); A(e->was_reset, uint8_t);
#undef A
    if (cfg->render) e->img = calloc((size_t)n * e->H * e->W * 3, 1);
    e->rowtab = calloc((size_t)e->H * 2, sizeof(float));
    e->pal = calloc((size_t)e->H * 4, sizeof(uint32_t));
    e->rowdepth = calloc((size_t)e->H, sizeof(float));
    if (cfg->render && cfg->depth) e->depth = calloc((size_t)n * e->H * e->W, sizeof(float));
    *out = e;
    return TRS_OK;
}

EXPORT int trso_destroy(trs_env* e)
{
    if (!e) return TRS_OK;
    free(e->px); free(e->py); free(e->pz); free(e->tang); free(e->start_yaw); free(e->map); free(e->rowtab); free(e->pal); free(e->rowdepth); free(e->depth); free(e->dpitch); free(e->sky);
    free(e->x); free(e->y); free(e->z); free(e->yaw); free(e->v); free(e->speed); free(e->cte); free(e->ep_return);
    free(e->last_return); free(e->steer_filt); free(e->seg_idx); free(e->ep_len); free(e->done); free(e->pending); free(e->was_reset); free(e->img); free(e->pre); free(e->mux);
    for (int k = 0; k < 32; ++k) free(e->scratch[k]);
    free(e);
    return TRS_OK;
}

EXPORT int trso_load_track(trs_env* e, const double* xyz, int np)
{
    if (!e || !xyz || np < 2) return fail(TRS_ERR_ARG, "bad track");
    free(e->px); free(e->py); free(e->pz); free(e->tang); free(e->start_yaw); free(e->map); e->map = NULL;
    e->np = np;
    e->px = malloc(sizeof(double) * np); e->py = malloc(sizeof(double) * np); e->pz = malloc(sizeof(double) * np);
    e->tang = malloc(sizeof(float) * 2 * np); e->start_yaw = malloc(sizeof(float) * np);
    for (int i = 0; i < np; ++i) { e->px[i] = xyz[3 * i]; e->py[i] = xyz[3 * i + 1]; e->pz[i] = xyz[3 * i + 2]; }
    int rc = build_track_tables(e);
    if (rc) return rc;
    for (int i = 0; i < e->n; ++i) {
        int gid = e->cfg.env_id_base + i;
        int si = (int)(((int64_t)TRS_START_STRIDE * gid) % np);
        e->x[i] = (float)e->px[si]; e->y[i] = (float)e->py[si]; e->z[i] = (float)e->pz[si];
        e->yaw[i] = e->start_yaw[si]; e->v[i] = 0.0f; e->speed[i] = 0.0f; e->cte[i] = 0.0f;
        e->seg_idx[i] = si; e->ep_return[i] = 0.0f; e->last_return[i] = 0.0f; e->ep_len[i] = 0;
        e->steer_filt[i] = 0.0f; e->done[i] = 0; e->pending[i] = 1;
    }
    e->step_count = 0;
    memset(e->stats, 0, sizeof e->stats);
    return TRS_OK;
}

EXPORT int trso_reset(trs_env* e, const uint8_t* mask)
{
    if (!e || !e->np) return fail(TRS_ERR_STATE, "no track loaded");
    for (int i = 0; i < e->n; ++i) if (!mask || mask[i]) e->pending[i] = 1;
    return TRS_OK;
}

static void preprocess_image(const trs_pre_config* c, const uint8_t* src, uint8_t* dst, int H, int W);
static void hsv_tables(void);

static int do_steps(struct trs_env* e, const float* st, const float* th, const float* br, const uint8_t* rs, int n_steps, int synth)
{
    if (!e || !e->np) return fail(TRS_ERR_STATE, "no track loaded");
    if (n_steps < 1) return fail(TRS_ERR_ARG, "n_steps < 1");
    if (!synth && (!st || !th)) return fail(TRS_ERR_ARG, "null controls");
    for (int k = 0; k < n_steps; ++k) {
        uint32_t t = (uint32_t)e->step_count;
#pragma omp parallel for schedule(static) num_threads(e->threads)
        for (int i = 0; i < e->n; ++i) {
            float steer, thr, brk = 0.0f, s, c;
            int reset_in = 0;
            if (synth) {
                /* the generator state advances only on steps that integrate; decide reset first */
                int will_reset = e->pending[i] || (e->cfg.auto_reset && e->done[i]);
                if (will_reset) { steer = 0.0f; thr = 0.0f; }
                else synth_controls(e->cfg.seed, (uint32_t)(e->cfg.env_id_base + i), t, &e->steer_filt[i], &steer, &thr);
            } else {
                steer = st[i]; thr = th[i]; brk = br ? br[i] : 0.0f; reset_in = (rs && k == 0) ? rs[i] != 0 : 0;
            }
            step_env(e, i, steer, thr, brk, reset_in, &s, &c);
            if (e->img) {
                render_env(e, i, s, c);
                if (e->has_frame_filter) {
                    /* the oracle takes the long way on purpose: render the raw frame, then run the full per-pixel filter
                     * over it -- the product filters the palette instead (trs_set_frame_filter) */
                    size_t fb = (size_t)e->H * e->W * 3;
                    uint8_t* tmp = malloc(fb);
                    preprocess_image(&e->frame_filter, e->img + (size_t)i * fb, tmp, e->H, e->W);
                    memcpy(e->img + (size_t)i * fb, tmp, fb);
                    free(tmp);
                }
            }
        }
        for (int i = 0; i < e->n; ++i) { e->stats[0] += e->done[i]; e->stats[1] += e->was_reset[i]; }
        e->step_count++;
    }
    return TRS_OK;
}

EXPORT int trso_step(trs_env* e, const float* st, const float* th, const float* br, const uint8_t* rs, int n)
{ return do_steps(e, st, th, br, rs, n, 0); }
EXPORT int trso_step_host(trs_env* e, const float* st, const float* th, const float* br, const uint8_t* rs, int n)
{ return do_steps(e, st, th, br, rs, n, 0); }
EXPORT int trso_step_sequence(trs_env* e, const float* st, const float* th, const float* br, const uint8_t* rs, int n_steps, int steps_per_launch)
{
    if (!e || !e->np) return fail(TRS_ERR_STATE, "no track loaded");
    if (n_steps < 1 || steps_per_launch < 1) return fail(TRS_ERR_ARG, "n_steps / steps_per_launch < 1");
    if (!st || !th) return fail(TRS_ERR_ARG, "null controls");
    for (int k = 0; k < n_steps; ++k) {                     /* one plain step per control set */
        size_t o = (size_t)k * (size_t)e->n;
        int rc = do_steps(e, st + o, th + o, br ? br + o : NULL, k == 0 ? rs : NULL, 1, 0);
        if (rc) return rc;
    }
    return TRS_OK;
}
EXPORT int trso_step_sequence_host(trs_env* e, const float* st, const float* th, const float* br, const uint8_t* rs, int n_steps, int steps_per_launch)
{ return trso_step_sequence(e, st, th, br, rs, n_steps, steps_per_launch); }

EXPORT int trso_step_synthetic(trs_env* e, int n_steps, int steps_per_launch)
{ (void)steps_per_launch; return do_steps(e, NULL, NULL, NULL, NULL, n_steps, 1); }

EXPORT int trso_get_state(trs_env* e, trs_state_view* o)
{
    if (!e || !o) return fail(TRS_ERR_ARG, "null argument");
    o->n_envs = e->n; o->img_h = e->H; o->img_w = e->W; o->n_points = e->np;
    o->img = e->img; o->pos_x = e->x; o->pos_y = e->y; o->pos_z = e->z; o->speed = e->speed; o->cte = e->cte;
    o->yaw = e->yaw; o->vel = e->v; o->seg_idx = e->seg_idx; o->ep_return = e->ep_return; o->last_return = e->last_return;
    o->ep_len = e->ep_len; o->done = e->done; o->step_count = e->step_count; o->depth = e->depth;
    return TRS_OK;
}

EXPORT int trso_copy_to_host(trs_env* e, int which, void* dst, size_t bytes)
{
    if (!e || !dst) return fail(TRS_ERR_ARG, "null argument");
    const void* src = NULL; size_t need = 0; size_t n = (size_t)e->n;
    switch (which) {
    case TRS_F_IMG: src = e->img; need = n * e->H * e->W * 3; break;
    case TRS_F_POS_X: src = e->x; need = n * 4; break;
    case TRS_F_POS_Y: src = e->y; need = n * 4; break;
    case TRS_F_POS_Z: src = e->z; need = n * 4; break;
    case TRS_F_SPEED: src = e->speed; need = n * 4; break;
    case TRS_F_CTE: src = e->cte; need = n * 4; break;
    case TRS_F_YAW: src = e->yaw; need = n * 4; break;
    case TRS_F_VEL: src = e->v; need = n * 4; break;
    case TRS_F_SEG_IDX: src = e->seg_idx; need = n * 4; break;
    case TRS_F_EP_RETURN: src = e->ep_return; need = n * 4; break;
    case TRS_F_LAST_RETURN: src = e->last_return; need = n * 4; break;
    case TRS_F_EP_LEN: src = e->ep_len; need = n * 4; break;
    case TRS_F_DONE: src = e->done; need = n; break;
    case TRS_F_MAP: src = e->map; need = (size_t)e->mi.map_words * e->mi.map_h * 4; break;
    case TRS_F_ROWTAB: src = e->rowtab; need = (size_t)e->H * 8; break;
    case TRS_F_PALETTE: src = e->pal; need = (size_t)e->H * 16; break;
    case TRS_F_TANGENT: src = e->tang; need = (size_t)e->np * 8; break;
    case TRS_F_STEER_FILT: src = e->steer_filt; need = n * 4; break;
    case TRS_F_STATS: src = e->stats; need = sizeof e->stats; break;
    case TRS_F_DEPTH: src = e->depth; need = n * e->H * e->W * 4; break;
    case TRS_F_ROWDEPTH: src = e->rowdepth; need = (size_t)e->H * 4; break;
    case TRS_F_DPITCH: src = e->dpitch; need = (size_t)e->np * 4; break;
    default: return fail(TRS_ERR_ARG, "unknown field");
    }
    if (!src) return fail(TRS_ERR_STATE, "field not available");
    if (bytes != need) return fail(TRS_ERR_ARG, "byte count mismatch");
    memcpy(dst, src, need);
    return TRS_OK;
}

EXPORT int trso_fetch_outputs(trs_env* e, uint8_t* img, float* x, float* y, float* z, float* speed, float* cte, int32_t* seg, uint8_t* done)
{
    if (!e || !e->np) return fail(TRS_ERR_STATE, "no track loaded");
    if (img && !e->img) return fail(TRS_ERR_STATE, "the env has no camera (cfg.render == 0)");
    size_t n = (size_t)e->n;
    if (img) memcpy(img, e->img, n * e->H * e->W * 3);
    if (x) memcpy(x, e->x, n * 4);
    if (y) memcpy(y, e->y, n * 4);
    if (z) memcpy(z, e->z, n * 4);
    if (speed) memcpy(speed, e->speed, n * 4);
    if (cte) memcpy(cte, e->cte, n * 4);
    if (seg) memcpy(seg, e->seg_idx, n * 4);
    if (done) memcpy(done, e->done, n);
    return TRS_OK;
}

EXPORT int trso_set_pose(trs_env* e, const float* x, const float* y, const float* z, const float* yaw, const float* v)
{
    if (!e || !e->np) return fail(TRS_ERR_STATE, "no track loaded");
    for (int i = 0; i < e->n; ++i) {
        if (x) e->x[i] = x[i];
        if (y) e->y[i] = y[i];
        if (z) e->z[i] = z[i];
        if (yaw) e->yaw[i] = yaw[i];
        if (v) e->v[i] = v[i];
        e->pending[i] = 0; e->done[i] = 0;
    }
    return TRS_OK;
}

EXPORT int trso_locate(trs_env* e, const double* xyz, int nq, int32_t* idx)
{
    if (!e || !e->np) return fail(TRS_ERR_STATE, "no track loaded");
    if (nq < 0 || (nq && (!xyz || !idx))) return fail(TRS_ERR_ARG, "bad query");
#pragma omp parallel for schedule(static) num_threads(e->threads)
    for (int q = 0; q < nq; ++q) idx[q] = l1_nearest(e, xyz[3 * q], xyz[3 * q + 1], xyz[3 * q + 2], NULL);
    return TRS_OK;
}

EXPORT int trso_map_info_get(trs_env* e, trs_map_info* o)
{
    if (!e || !o || !e->np) return fail(TRS_ERR_STATE, "no track loaded");
    *o = e->mi; o->lds_bytes = 0;
    return TRS_OK;
}

EXPORT int trso_sync(trs_env* e) { (void)e; return TRS_OK; }
/* the multi-GPU exchange (include/trsim.h): the oracle has no transport, so its "communicator" is the one-rank case, where the
 * all-gather is a copy; more ranks are gathered by the tests' own channel (gloo) */
EXPORT int trso_comm_get_unique_id(void* id) { if (!id) return TRS_ERR_ARG; memset(id, 0, 128); return TRS_OK; }
EXPORT int trso_comm_init(trs_env* e, int rank, int world, const void* id) { (void)id; if (!e || world != 1 || rank != 0) return TRS_ERR_ARG; e->comm_ready = 1; return TRS_OK; }
EXPORT int trso_comm_destroy(trs_env* e) { if (!e) return TRS_ERR_ARG; e->comm_ready = 0; return TRS_OK; }
EXPORT int trso_allgather_returns(trs_env* e, const float** d_out, float* h_out)
{
    if (!e) return TRS_ERR_ARG;
    if (!e->comm_ready) return TRS_ERR_STATE;
    if (d_out) *d_out = e->ep_return;
    if (h_out) memcpy(h_out, e->ep_return, (size_t)e->n * sizeof(float));
    return TRS_OK;
}
/* "device" buffers of a CPU env are host buffers */
EXPORT int trso_scratch(trs_env* e, int slot, size_t bytes, void** out)
{
    if (!e || !out || slot < 0 || slot >= 32) return TRS_ERR_ARG;
    if (bytes > e->scratch_bytes[slot]) {
        free(e->scratch[slot]);
        e->scratch[slot] = calloc(1, bytes < 256 ? 256 : bytes);
        if (!e->scratch[slot]) { e->scratch_bytes[slot] = 0; return TRS_ERR_NOMEM; }
        e->scratch_bytes[slot] = bytes < 256 ? 256 : bytes;
    }
    *out = e->scratch[slot];
    return TRS_OK;
}
EXPORT int trso_upload(trs_env* e, void* dst, const void* src, size_t bytes) { if (!e || (bytes && (!dst || !src))) return TRS_ERR_ARG; memcpy(dst, src, bytes); return TRS_OK; }
EXPORT int trso_counters(trs_env* e, uint64_t out[4]) { if (!e || !out) return TRS_ERR_ARG; out[0] = out[1] = out[3] = 0; out[2] = e->step_count; return TRS_OK; }
/* there are no streams on the CPU */
EXPORT int trso_stream_wait_external(trs_env* e, void* s) { (void)s; return e ? TRS_OK : TRS_ERR_ARG; }
EXPORT int trso_stream_signal_external(trs_env* e, void* s) { (void)s; return e ? TRS_OK : TRS_ERR_ARG; }
/* how steps reach the GPU (include/trsim.h: trs_set_step_mode) changes no result: accepted and ignored here */
EXPORT int trso_step_wait(trs_env* e, const float* st, const float* th, const float* br, const uint8_t* rs, int n) { return trso_step(e, st, th, br, rs, n); }   /* synchronous anyway */
EXPORT int trso_quiesce(trs_env* e) { return e ? TRS_OK : TRS_ERR_ARG; }   /* nothing is ever resident here */
EXPORT int trso_set_step_mode(trs_env* e, int mode, int idle_us) { (void)idle_us; if (!e) return TRS_ERR_ARG; return (mode == 0 || mode == 1) ? TRS_OK : TRS_ERR_ARG; }
EXPORT int trso_get_step_mode(trs_env* e, int* mode, int* fell_back) { if (!e) return TRS_ERR_ARG; if (mode) *mode = 0; if (fell_back) *fell_back = 0; return TRS_OK; }
EXPORT int trso_event_record(trs_env* e, int slot) { (void)e; (void)slot; return TRS_OK; }
EXPORT int trso_event_elapsed_ms(trs_env* e, int a, int b, float* ms) { (void)e; (void)a; (void)b; if (ms) *ms = 0.0f; return TRS_OK; }
EXPORT int trso_device_count(int* out) { if (out) *out = 0; return TRS_OK; }
EXPORT const char* trso_last_error(void) { return g_err; }


/* ------------------------------------------------------------------ image path (rows a10, a11 colour masks, a13)
 * Restates ImgPreprocessing.__trim_brightness_contrast / __color_filter / __merge
 * (/root/reference/TritonRacerSim/components/img_preprocessing.py:57-74,81-102).  The numpy part (:92-99) is
 * restated operation by operation in binary32 and is PINNED by tests/test_image_path.py against numpy itself;
 * cv2.mean (:88) is restated as exact integer sums divided in binary64; cv2.cvtColor(RGB2HSV) + cv2.inRange
 * (:66,:71) follow OpenCV's published 8-bit algorithm (fixed-point tables, H in [0,180)) — cv2 is not
 * installed here, so that part is PARITY UNPINNED. */

static int g_sdiv[256], g_hdiv[256], g_tables_ready;

static void hsv_tables(void)
{
    if (g_tables_ready) return;
    g_sdiv[0] = g_hdiv[0] = 0;
    for (int i = 1; i < 256; ++i) {
        g_sdiv[i] = (int)lrint((255 << 12) / (1.0 * i));
        g_hdiv[i] = (int)lrint((180 << 12) / (6.0 * i));
    }
    g_tables_ready = 1;
}

static void rgb2hsv_u8(int r, int g, int b, int* ho, int* so, int* vo)
{
    int v = r > g ? r : g; if (b > v) v = b;
    int vmin = r < g ? r : g; if (b < vmin) vmin = b;
    int diff = v - vmin;
    int vr = (v == r) ? -1 : 0, vg = (v == g) ? -1 : 0;
    int sat = (diff * g_sdiv[v] + (1 << 11)) >> 12;
    int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    h = (h * g_hdiv[diff] + (1 << 11)) >> 12;
    if (h < 0) h += 180;
    *ho = h > 255 ? 255 : h; *so = sat > 255 ? 255 : sat; *vo = v;
}


/* cv2.Canny(img, a, b) on an 8-bit 3-channel image, aperture 3, L1 gradient — OpenCV's published algorithm: per-channel
 * 3x3 Sobel with replicated borders, the channel with the largest |dx|+|dy| wins (first on ties), non-maximum suppression
 * in three direction classes with the fixed-point tangent test (tan 22.5 deg = 13573 / 2^15), magnitudes outside the image
 * are 0, 8-connected hysteresis between floor(min(a,b)) and floor(max(a,b)).  PARITY UNPINNED (cv2 is not installed). */
static void canny_u8c3(const uint8_t* img, int H, int W, int ta, int tb, uint8_t* edge)
{
    const int low = ta < tb ? ta : tb, high = ta < tb ? tb : ta;
    const int MW = W + 2;
    int* mag = calloc((size_t)(H + 2) * MW, sizeof(int));
    short* gx = malloc(sizeof(short) * (size_t)H * W), *gy = malloc(sizeof(short) * (size_t)H * W);
    uint8_t* map = malloc((size_t)H * W);
    int* stack = malloc(sizeof(int) * (size_t)H * W);
    int sp = 0;
#define PX(y, x, c) ((int)img[(((size_t)((y) < 0 ? 0 : ((y) >= H ? H - 1 : (y)))) * W + ((x) < 0 ? 0 : ((x) >= W ? W - 1 : (x)))) * 3 + (c)])
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int bn = -1, bdx = 0, bdy = 0;
            for (int c = 0; c < 3; ++c) {
                int dx = (PX(y - 1, x + 1, c) + 2 * PX(y, x + 1, c) + PX(y + 1, x + 1, c)) - (PX(y - 1, x - 1, c) + 2 * PX(y, x - 1, c) + PX(y + 1, x - 1, c));
                int dy = (PX(y + 1, x - 1, c) + 2 * PX(y + 1, x, c) + PX(y + 1, x + 1, c)) - (PX(y - 1, x - 1, c) + 2 * PX(y - 1, x, c) + PX(y - 1, x + 1, c));
                int nrm = abs(dx) + abs(dy);
                if (nrm > bn) { bn = nrm; bdx = dx; bdy = dy; }
            }
            mag[(size_t)(y + 1) * MW + x + 1] = bn; gx[(size_t)y * W + x] = (short)bdx; gy[(size_t)y * W + x] = (short)bdy;
        }
#undef PX
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int* m0 = mag + (size_t)(y + 1) * MW + x + 1;
            const int m = *m0;
            int ismax = 0;
            if (m > low) {
                const int xs = gx[(size_t)y * W + x], ys = gy[(size_t)y * W + x];
                const int ax = abs(xs), ay = abs(ys) << 15;
                const int tg22x = ax * 13573;
                if (ay < tg22x) ismax = m > m0[-1] && m >= m0[1];
                else {
                    const int tg67x = tg22x + (ax << 16);
                    if (ay > tg67x) ismax = m > m0[-MW] && m >= m0[MW];
                    else { const int s = (xs ^ ys) < 0 ? -1 : 1; ismax = m > m0[-MW - s] && m > m0[MW + s]; }
                }
            }
            uint8_t v = 1;
            if (ismax) { if (m > high) { v = 2; stack[sp++] = y * W + x; } else v = 0; }
            map[(size_t)y * W + x] = v;
        }
    while (sp) {
        const int p = stack[--sp], y = p / W, x = p % W;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = y + dy, xx = x + dx;
                if ((dy || dx) && yy >= 0 && yy < H && xx >= 0 && xx < W && map[(size_t)yy * W + xx] == 0) { map[(size_t)yy * W + xx] = 2; stack[sp++] = yy * W + xx; }
            }
    }
    for (size_t i = 0; i < (size_t)H * W; ++i) edge[i] = map[i] == 2 ? 255 : 0;
    free(mag); free(gx); free(gy); free(map); free(stack);
}

static void preprocess_image(const trs_pre_config* c, const uint8_t* src, uint8_t* dst, int H, int W)
{
    uint8_t* trimmed = c->edge_detection_enabled ? malloc((size_t)H * W * 3) : NULL;
    int r0 = 40 < H ? 40 : H, r1 = 119 < H ? 119 : H;
    uint64_t sum[3] = {0, 0, 0};
    for (int y = r0; y < r1; ++y)
        for (int x = 0; x < W; ++x)
            for (int ch = 0; ch < 3; ++ch) sum[ch] += src[((size_t)y * W + x) * 3 + ch];
    double cnt = (double)(r1 - r0) * (double)W;
    double cur = 0.0;
    for (int ch = 0; ch < 3; ++ch) cur = cur + (cnt > 0 ? (double)sum[ch] / cnt : 0.0);
    cur = cur + 0.0;                                        /* cv2.mean returns a 4-tuple; the 4th entry is 0 */
    double delta = (c->brightness_baseline - cur) / 3;
    float deltaf = (float)delta, off = c->contrast_offset, con = c->contrast_ratio;
    for (size_t i = 0; i < (size_t)H * W; ++i) {
        int t[3];
        for (int ch = 0; ch < 3; ++ch) {
            float x = (float)src[3 * i + ch];
            if (c->dynamic_brightness) x = x + deltaf;
            x = x - off;
            x = x * con;
            x = x + off;
            x = x < 0.0f ? 0.0f : (x > 255.0f ? 255.0f : x);
            t[ch] = (int)x;                                 /* astype(uint8): truncation */
        }
        int o[3] = {t[0], t[1], t[2]};
        if (trimmed) { trimmed[3 * i] = (uint8_t)t[0]; trimmed[3 * i + 1] = (uint8_t)t[1]; trimmed[3 * i + 2] = (uint8_t)t[2]; }
        if (c->color_filter_enabled) {
            int h, sa, v;
            rgb2hsv_u8(t[0], t[1], t[2], &h, &sa, &v);
            for (int f = 0; f < c->n_filters; ++f) {
                int in = h >= c->hsv_lo[f][0] && h <= c->hsv_hi[f][0] && sa >= c->hsv_lo[f][1] && sa <= c->hsv_hi[f][1] &&
                         v >= c->hsv_lo[f][2] && v <= c->hsv_hi[f][2];
                o[c->dst_channel[f]] = in ? 255 : 0;
            }
        }
        dst[3 * i] = (uint8_t)o[0]; dst[3 * i + 1] = (uint8_t)o[1]; dst[3 * i + 2] = (uint8_t)o[2];
    }
    if (trimmed) {      /* the edge layer is computed on the trimmed image and merged last (img_preprocessing.py:48-53) */
        uint8_t* edge = malloc((size_t)H * W);
        canny_u8c3(trimmed, H, W, c->edge_threshold_a, c->edge_threshold_b, edge);
        for (size_t i = 0; i < (size_t)H * W; ++i) dst[3 * i + c->edge_dst_channel] = edge[i];
        free(edge); free(trimmed);
    }
}

static int check_pre(const trs_pre_config* c)
{
    if (!c || c->struct_size != sizeof(trs_pre_config)) return fail(TRS_ERR_ARG, "trs_pre_config.struct_size mismatch");
    if (c->edge_detection_enabled && (c->edge_dst_channel < 0 || c->edge_dst_channel > 2)) return fail(TRS_ERR_ARG, "edge_dst_channel out of range");
    if (c->n_filters < 0 || c->n_filters > 4) return fail(TRS_ERR_ARG, "n_filters out of range");
    for (int f = 0; f < c->n_filters; ++f) if (c->dst_channel[f] < 0 || c->dst_channel[f] > 2) return fail(TRS_ERR_ARG, "dst_channel out of range");
    return TRS_OK;
}

EXPORT void trso_default_pre_config(trs_pre_config* c)
{
    memset(c, 0, sizeof *c);
    c->struct_size = (uint32_t)sizeof *c;
    c->brightness_baseline = 550.0; c->contrast_ratio = 1.0f; c->contrast_offset = 125.0f;
    c->n_filters = 2;                                       /* core/config.py:23-24: white and yellow */
    const uint8_t lo[2][3] = {{0, 0, 130}, {25, 180, 155}}, hi[2][3] = {{180, 64, 255}, {43, 255, 255}};
    memcpy(c->hsv_lo, lo, sizeof lo); memcpy(c->hsv_hi, hi, sizeof hi);
    c->dst_channel[0] = 0; c->dst_channel[1] = 1;
    c->edge_threshold_a = 60; c->edge_threshold_b = 100; c->edge_dst_channel = 2;
}

EXPORT int trso_set_frame_filter(trs_env* e, const trs_pre_config* c)
{
    if (!e) return fail(TRS_ERR_ARG, "null handle");
    if (!e->cfg.render) return fail(TRS_ERR_STATE, "the env has no camera (cfg.render == 0)");
    if (!c) { e->has_frame_filter = 0; return TRS_OK; }
    int rc = check_pre(c);
    if (rc) return rc;
    if (c->edge_detection_enabled) return fail(TRS_ERR_ARG, "the Canny layer is a neighbourhood operator: not a palette filter, use trs_preprocess");
    hsv_tables();
    e->frame_filter = *c; e->has_frame_filter = 1;
    return TRS_OK;
}

EXPORT int trso_preprocess_host(trs_env* e, const trs_pre_config* c, const uint8_t* src, uint8_t* dst, int n)
{
    if (!e || !src || !dst || n < 0) return fail(TRS_ERR_ARG, "bad argument");
    int rc = check_pre(c);
    if (rc) return rc;
    hsv_tables();
    size_t stride = (size_t)e->H * e->W * 3;
#pragma omp parallel for schedule(static) num_threads(e->threads)
    for (int i = 0; i < n; ++i) preprocess_image(c, src + i * stride, dst + i * stride, e->H, e->W);
    return TRS_OK;
}

EXPORT int trso_preprocess(trs_env* e, const trs_pre_config* c, const uint8_t* src, uint8_t* dst, int n, const uint8_t** out)
{
    if (!e) return fail(TRS_ERR_ARG, "null handle");
    if (!src) { if (!e->img || n != e->n) return fail(TRS_ERR_ARG, "latest-frame source needs n_images == n_envs and a camera"); src = e->img; }
    if (!dst) { if (!e->pre) e->pre = malloc((size_t)e->n * e->H * e->W * 3); if (n > e->n) return fail(TRS_ERR_ARG, "own buffer holds n_envs frames"); dst = e->pre; }
    if (out) *out = dst;
    return trso_preprocess_host(e, c, src, dst, n);
}

EXPORT int trso_normalize_host(trs_env* e, const uint8_t* src, float* dst, int n)
{
    if (!e || !src || !dst || n < 0) return fail(TRS_ERR_ARG, "bad argument");
    size_t total = (size_t)n * e->H * e->W * 3;
    for (size_t i = 0; i < total; ++i) dst[i] = (float)src[i] / 255.0f;     /* keras_pilot.py:49-50 */
    return TRS_OK;
}

EXPORT int trso_normalize(trs_env* e, const uint8_t* src, float* dst, int n)
{
    if (!e) return fail(TRS_ERR_ARG, "null handle");
    if (!src) { if (!e->img || n != e->n) return fail(TRS_ERR_ARG, "latest-frame source needs n_images == n_envs and a camera"); src = e->img; }
    return trso_normalize_host(e, src, dst, n);
}

/* DriverAssistance.step restated (/root/reference/TritonRacerSim/components/driver_assistance.py:13-31); PINNED by
 * tests/golden/driver_assistance.json (captured from the reference). */
EXPORT int trso_driver_assist_host(trs_env* e, int mode, double k, float* st, float* th, float* br, const float* sp, int n)
{
    if (!e || !st || !th || !br || !sp || n < 0 || (mode != 0 && mode != 1)) return fail(TRS_ERR_ARG, "bad argument");
    for (int i = 0; i < n; ++i) {
        double steering = st[i], throttle = th[i], breaking = br[i], speed = sp[i];
        if (mode == 0 && speed != 0) {
            double max_steering = k / speed;
            if (steering > max_steering) { steering = max_steering; throttle = -0.1; }
            else if (steering < max_steering * -1) { steering = max_steering * -1; throttle = -0.1; }
        } else if (mode == 1 && steering != 0) {
            double max_speed = k / steering;
            if (speed > max_speed) { throttle = 0.0; breaking = 0.0; }
        }
        st[i] = (float)steering; th[i] = (float)throttle; br[i] = (float)breaking;
    }
    return TRS_OK;
}
EXPORT int trso_driver_assist(trs_env* e, int mode, double k, float* st, float* th, float* br, const float* sp, int n)
{ if (e && !sp) sp = e->speed; return trso_driver_assist_host(e, mode, k, st, th, br, sp, n); }

/* ControlMultiplexer.step restated (/root/reference/TritonRacerSim/components/controlmultiplexer.py:24-70) on the env's
 * fixed tick: a lock-end thread's sleep of `duration` seconds becomes an END EVENT `ticks` ticks after its trigger.
 * Parity UNPINNED by the reference (controlmultiplexer.py imports pygame through controller.py, absent here; the
 * reference holds no test for it); checked against the independent event-queue model in oracle/pyref.py. */
#define MUX_PENDING 8
struct mux_car {
    int last_mode;                       /* :10 DriveMode.HUMAN */
    int steering_lock_active, throttle_lock_active;   /* :12,:17 */
    int n_trig;                          /* triggers seen; the last MUX_PENDING are kept */
    int trig_tick[MUX_PENDING];
};

EXPORT void trso_default_mux_config(trs_mux_config* c)
{
    if (!c) return;
    memset(c, 0, sizeof(*c));
    c->struct_size = sizeof(*c);
    c->throttle_lock_enabled = 0; c->throttle_lock_value = 1.0f; c->throttle_lock_ticks = 100;
    c->steering_lock_enabled = 0; c->steering_lock_value = 0.0f; c->steering_lock_ticks = 60;
}

EXPORT int trso_control_mux_reset(trs_env* e)
{
    if (!e) return fail(TRS_ERR_ARG, "null handle");
    free(e->mux);
    e->mux = calloc((size_t)e->n, sizeof(struct mux_car));
    if (!e->mux) return fail(TRS_ERR_DEVICE, "out of memory");
    for (int i = 0; i < e->n; ++i) e->mux[i].last_mode = TRS_MODE_HUMAN;
    e->mux_tick = 0;
    return TRS_OK;
}

EXPORT int trso_control_mux_host(trs_env* e, const trs_mux_config* c, const uint8_t* mode, const float* us, const float* ut, const float* ub,
                                 const float* as, const float* at, const float* ab, float* os, float* ot, float* ob, int n)
{
    if (!e || !c || !mode || !us || !ut || !ub || !as || !at || !ab || !os || !ot || !ob) return fail(TRS_ERR_ARG, "null argument");
    if (c->struct_size != sizeof(trs_mux_config)) return fail(TRS_ERR_ARG, "trs_mux_config.struct_size mismatch");
    if (n < 0 || n > e->n) return fail(TRS_ERR_ARG, "n must be in [0, n_envs] (the lock state is kept per env)");
    if ((c->throttle_lock_enabled && c->throttle_lock_ticks < 1) || (c->steering_lock_enabled && c->steering_lock_ticks < 1))
        return fail(TRS_ERR_ARG, "lock ticks must be >= 1");
    if (!e->mux) { int rc = trso_control_mux_reset(e); if (rc) return rc; }
    const int now = e->mux_tick;
    for (int i = 0; i < n; ++i) {
        struct mux_car* m = &e->mux[i];
        /* __end_throttle_lock / __end_steering_lock (:51-54, :67-70): one per trigger, each clears the flag when it wakes */
        int kept = m->n_trig < MUX_PENDING ? m->n_trig : MUX_PENDING;
        for (int q = 0; q < kept; ++q) {
            int t0 = m->trig_tick[q];
            if (c->throttle_lock_enabled && now - t0 == c->throttle_lock_ticks) m->throttle_lock_active = 0;
            if (c->steering_lock_enabled && now - t0 == c->steering_lock_ticks) m->steering_lock_active = 0;
        }
        float r0 = 0, r1 = 0, r2 = 0; int have = 1;
        switch (mode[i]) {
        case TRS_MODE_HUMAN:       r0 = us[i]; r1 = ut[i]; r2 = ub[i]; break;    /* :26-27 */
        case TRS_MODE_AI_STEERING: r0 = as[i]; r1 = ut[i]; r2 = ub[i]; break;    /* :28-29 */
        case TRS_MODE_AI:          r0 = as[i]; r1 = at[i]; r2 = ab[i]; break;    /* :30-31 */
        default: have = 0;
        }
        if (m->last_mode != TRS_MODE_AI && mode[i] == TRS_MODE_AI) {             /* :33 */
            int started = 0;
            if (c->throttle_lock_enabled) { m->throttle_lock_active = 1; started = 1; }   /* :45-49 */
            if (c->steering_lock_enabled) { m->steering_lock_active = 1; started = 1; }   /* :60-65 */
            if (started) { m->trig_tick[m->n_trig % MUX_PENDING] = now; m->n_trig++; }
        }
        if (have) {
            if (m->steering_lock_active) r0 = c->steering_lock_value;            /* :37-38 */
            if (m->throttle_lock_active) r1 = c->throttle_lock_value;            /* :39-40 */
            os[i] = r0; ot[i] = r1; ob[i] = r2;
        }
        m->last_mode = mode[i];                                                  /* :42 */
    }
    e->mux_tick = now + 1;
    return TRS_OK;
}
EXPORT int trso_control_mux(trs_env* e, const trs_mux_config* c, const uint8_t* mode, const float* us, const float* ut, const float* ub,
                            const float* as, const float* at, const float* ab, float* os, float* ot, float* ob, int n)
{ return trso_control_mux_host(e, c, mode, us, ut, ub, as, at, ab, os, ot, ob, n); }

/* oracle-only: number of OpenMP threads used by step / locate (cpu_baseline "cores") */
EXPORT int trso_set_threads(trs_env* e, int n)
{
    if (!e || n < 1) return fail(TRS_ERR_ARG, "bad thread count");
#ifdef _OPENMP
    int mx = omp_get_num_procs();
    e->threads = n > mx ? mx : n;
#else
    e->threads = 1;
#endif
    return e->threads;
}
