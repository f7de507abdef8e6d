// trsim_tables.cpp — see trsim_tables.hpp.  Product code: independent of oracle/.
#include "trsim_tables.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <limits>

#include "../../include/trsim_spec.h"

namespace trsim {
namespace {

struct Pt { double x, z; };

// Unit tangent at raw sample i: from the previous to the next sample that differs from it in (x, z);
// the recorded lap is treated as a closed loop.  (Spec: "cte" and start heading.)
bool tangent_at(const TrackTables& t, int i, double& tx, double& tz)
{
    const int n = t.n_points;
    auto differs = [&](int k) { return t.px[k] != t.px[i] || t.pz[k] != t.pz[i]; };
    int nxt = -1, prv = -1;
    for (int step = 1; step <= n; ++step) { int k = (i + step) % n; if (differs(k)) { nxt = k; break; } }
    if (nxt < 0) return false;
    for (int step = 1; step <= n; ++step) { int k = (i + n - step % n) % n; if (differs(k)) { prv = k; break; } }
    tx = t.px[nxt] - t.px[prv];
    tz = t.pz[nxt] - t.pz[prv];
    double len = std::sqrt(tx * tx + tz * tz);
    if (len == 0.0) {
        tx = t.px[nxt] - t.px[i];
        tz = t.pz[nxt] - t.pz[i];
        len = std::sqrt(tx * tx + tz * tz);
    }
    tx /= len;
    tz /= len;
    return true;
}

std::vector<Pt> dedup_closed(const TrackTables& t)
{
    std::vector<Pt> q;
    q.reserve(t.n_points);
    for (int i = 0; i < t.n_points; ++i) {
        if (!q.empty() && q.back().x == t.px[i] && q.back().z == t.pz[i]) continue;
        q.push_back({t.px[i], t.pz[i]});
    }
    if (q.size() > 1 && q.back().x == q.front().x && q.back().z == q.front().z) q.pop_back();
    return q;
}

int round_colour(double v) { return (int)std::floor(v + 0.5); }

}  // namespace

int build_tables(const trs_config& cfg, const double* xyz, int n_points, TrackTables& T, std::string& err)
{
    if (!xyz || n_points < 2) { err = "bad track"; return TRS_ERR_ARG; }
    T = TrackTables{};
    T.n_points = n_points;
    T.px.resize(n_points); T.py.resize(n_points); T.pz.resize(n_points);
    for (int i = 0; i < n_points; ++i) { T.px[i] = xyz[3 * i]; T.py[i] = xyz[3 * i + 1]; T.pz[i] = xyz[3 * i + 2]; }

    T.tangent.resize(2 * (size_t)n_points);
    T.start_yaw.resize(n_points);
    for (int i = 0; i < n_points; ++i) {
        double tx, tz;
        if (!tangent_at(T, i, tx, tz)) { err = "track has no two distinct points"; return TRS_ERR_ARG; }
        T.tangent[2 * i] = (float)tx;
        T.tangent[2 * i + 1] = (float)tz;
        T.start_yaw[i] = (float)std::atan2(tx, tz);
    }

    // ---- nearest-point accelerator grid (exactness rule: trsim_spec.h R3) ---------------------------
    {
        double gx0 = T.px[0], gx1 = T.px[0], gz0 = T.pz[0], gz1 = T.pz[0];
        for (int i = 1; i < n_points; ++i) {
            gx0 = std::min(gx0, T.px[i]); gx1 = std::max(gx1, T.px[i]);
            gz0 = std::min(gz0, T.pz[i]); gz1 = std::max(gz1, T.pz[i]);
        }
        const double g = TRS_NEAR_GRID_CELL;
        T.grid_x0 = gx0; T.grid_z0 = gz0;
        T.grid_nx = (int)std::floor((gx1 - gx0) / g) + 1;
        T.grid_nz = (int)std::floor((gz1 - gz0) / g) + 1;
        if (n_points > 65535 || (long long)T.grid_nx * T.grid_nz > 60000) { T.grid_nx = T.grid_nz = 0; }   // accelerator off: full scans only
        else {
            const int ncell = T.grid_nx * T.grid_nz;
            std::vector<int> cell(n_points), count(ncell + 1, 0);
            for (int i = 0; i < n_points; ++i) {
                const int cx = (int)std::floor((T.px[i] - gx0) / g), cz = (int)std::floor((T.pz[i] - gz0) / g);
                cell[i] = cz * T.grid_nx + cx;
                ++count[cell[i] + 1];
            }
            for (int c = 0; c < ncell; ++c) count[c + 1] += count[c];
            T.grid_start.assign(count.begin(), count.end());
            T.grid_pts.assign(n_points, 0);
            std::vector<int> fill(count.begin(), count.end() - 1);
            for (int i = 0; i < n_points; ++i) T.grid_pts[fill[cell[i]]++] = (uint16_t)i;
        }
    }

    // ---- surface-class map -------------------------------------------------------------------
    const std::vector<Pt> line = dedup_closed(T);
    const int m = (int)line.size();
    double xmin = line[0].x, xmax = line[0].x, zmin = line[0].z, zmax = line[0].z;
    for (const Pt& p : line) {
        xmin = std::min(xmin, p.x); xmax = std::max(xmax, p.x);
        zmin = std::min(zmin, p.z); zmax = std::max(zmax, p.z);
    }
    double cell = TRS_MAP_CELL_MIN, x0 = 0, z0 = 0;
    int gw = 0, gh = 0, mw = 0;
    while (true) {
        x0 = std::floor((xmin - cfg.map_margin) / cell) * cell;
        z0 = std::floor((zmin - cfg.map_margin) / cell) * cell;
        gw = (int)std::ceil((xmax + cfg.map_margin - x0) / cell);
        gh = (int)std::ceil((zmax + cfg.map_margin - z0) / cell);
        mw = (gw + 15) / 16;
        if ((size_t)mw * 4 * (size_t)gh <= (size_t)TRS_MAP_LDS_BUDGET) break;
        cell *= 2.0;
        if (cell > 64.0) { err = "track too large for the map budget"; return TRS_ERR_LIMIT; }
    }
    T.info.map_w = gw; T.info.map_h = gh; T.info.map_words = mw;
    T.info.cell = cell; T.info.x0 = x0; T.info.z0 = z0; T.info.n_points = n_points;
    T.map_x0f = (float)x0; T.map_z0f = (float)z0; T.inv_cellf = (float)(1.0 / cell);

    const size_t ncell = (size_t)gw * gh;
    std::vector<double> dist2(ncell, std::numeric_limits<double>::infinity());
    std::vector<double> arc(ncell, 0.0);
    const double reach = cfg.road_half + cfg.edge_half + cell;
    double s_acc = 0.0;
    for (int k = 0; k < m; ++k) {
        const Pt a = line[k], b = line[(k + 1) % m];
        const double abx = b.x - a.x, abz = b.z - a.z;
        const double len2 = abx * abx + abz * abz;
        const double len = std::sqrt(len2);
        int ix0 = (int)std::floor((std::min(a.x, b.x) - reach - x0) / cell);
        int ix1 = (int)std::floor((std::max(a.x, b.x) + reach - x0) / cell);
        int iz0 = (int)std::floor((std::min(a.z, b.z) - reach - z0) / cell);
        int iz1 = (int)std::floor((std::max(a.z, b.z) + reach - z0) / cell);
        ix0 = std::max(ix0, 0); iz0 = std::max(iz0, 0);
        ix1 = std::min(ix1, gw - 1); iz1 = std::min(iz1, gh - 1);
        for (int iz = iz0; iz <= iz1; ++iz) {
            const double cz = z0 + ((double)iz + 0.5) * cell;
            for (int ix = ix0; ix <= ix1; ++ix) {
                const double cx = x0 + ((double)ix + 0.5) * cell;
                const double apx = cx - a.x, apz = cz - a.z;
                double t = (apx * abx + apz * abz) / len2;
                t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
                const double ex = cx - (a.x + t * abx), ez = cz - (a.z + t * abz);
                const double d2 = ex * ex + ez * ez;
                const size_t ci = (size_t)iz * gw + ix;
                if (d2 < dist2[ci]) { dist2[ci] = d2; arc[ci] = s_acc + t * len; }
            }
        }
        s_acc += len;
    }
    T.map.assign((size_t)mw * gh, 0u);
    for (int iz = 1; iz < gh - 1; ++iz) {
        for (int ix = 1; ix < gw - 1; ++ix) {
            const size_t ci = (size_t)iz * gw + ix;
            if (!(dist2[ci] < std::numeric_limits<double>::infinity())) continue;
            const double d = std::sqrt(dist2[ci]);
            uint32_t cls = TRS_CLS_GRASS;
            if (d <= cfg.centre_half && std::fmod(arc[ci], cfg.dash_period) < cfg.dash_on) cls = TRS_CLS_CENTRE;
            else if (std::fabs(d - cfg.road_half) <= cfg.edge_half) cls = TRS_CLS_EDGE;
            else if (d < cfg.road_half) cls = TRS_CLS_ROAD;
            T.map[(size_t)iz * mw + (ix >> 4)] |= cls << ((ix & 15) * 2);
        }
    }

    // ---- camera rows + palette ---------------------------------------------------------------
    const std::array<std::array<int, 3>, 4> base = {{TRS_RGB_GRASS, TRS_RGB_ROAD, TRS_RGB_EDGE, TRS_RGB_CENTRE}};
    const std::array<int, 3> fog = TRS_RGB_FOG, sky_top = TRS_RGB_SKY_TOP, sky_hor = TRS_RGB_SKY_HOR;
    const int H = cfg.img_h;
    const double kPi = 3.14159265358979323846;
    const double half_h = (double)H / 2.0;
    const double f = half_h / std::tan(cfg.fov_v_deg * kPi / 180.0 / 2.0);
    const double pitch = cfg.cam_pitch_deg * kPi / 180.0;
    const double cp = std::cos(pitch), sp = std::sin(pitch);
    T.rowtab.assign(2 * (size_t)H, 0.0f);
    T.palette.assign(4 * (size_t)H, 0u);
    T.rowdepth.assign((size_t)H, (float)cfg.z_far);
    for (int v = 0; v < H; ++v) {
        const double yn = (half_h - ((double)v + 0.5)) / f;
        const double dy = yn * cp - sp, dz = yn * sp + cp;
        std::array<std::array<int, 3>, 4> rgb{};
        if (dy >= -1e-6) {                                   // sky
            double g = ((double)v + 0.5) / half_h;
            if (g > 1.0) g = 1.0;
            for (int ch = 0; ch < 3; ++ch) {
                const int val = round_colour((double)sky_top[ch] + ((double)sky_hor[ch] - (double)sky_top[ch]) * g);
                for (auto& c : rgb) c[ch] = val;
            }
        } else {
            const double t = cfg.cam_h / (-dy);
            const double fwd = t * dz;
            if (fwd > cfg.z_far) {                           // beyond the far plane: fogged grass
                for (int ch = 0; ch < 3; ++ch) {
                    const int val = round_colour((double)base[0][ch] * (1.0 - TRS_FOG_MAX) + (double)fog[ch] * TRS_FOG_MAX);
                    for (auto& c : rgb) c[ch] = val;
                }
            } else {
                T.rowtab[2 * v] = (float)(fwd / cell);
                T.rowtab[2 * v + 1] = (float)((t / f) / cell);
                T.rowdepth[v] = (float)fwd;
                const double fw = TRS_FOG_MAX * (fwd / cfg.z_far);
                for (int c = 0; c < 4; ++c)
                    for (int ch = 0; ch < 3; ++ch)
                        rgb[c][ch] = round_colour((double)base[c][ch] * (1.0 - fw) + (double)fog[ch] * fw);
            }
        }
        for (int c = 0; c < 4; ++c)
            T.palette[4 * v + c] = (uint32_t)rgb[c][0] | ((uint32_t)rgb[c][1] << 8) | ((uint32_t)rgb[c][2] << 16);
    }

    // ---- tracks with elevation (include/trsim_spec.h): is the track hilly, and by how much does the road A samples ahead tilt against the road here ----
    {
        const int n = n_points;
        const auto mm = std::minmax_element(T.py.begin(), T.py.end());
        T.hills = (*mm.second - *mm.first) > TRS_HILL_MIN_RANGE;
        T.dpitch.assign((size_t)n, 0.0f);
        if (T.hills) {
            std::vector<double> step((size_t)n), slope((size_t)n);
            auto wrap = [n](int i) { return ((i % n) + n) % n; };
            for (int i = 0; i < n; ++i) {
                const int j = wrap(i + 1);
                step[i] = std::sqrt((T.px[j] - T.px[i]) * (T.px[j] - T.px[i]) + (T.pz[j] - T.pz[i]) * (T.pz[j] - T.pz[i]));
            }
            for (int i = 0; i < n; ++i) {
                double run = 0.0;                                          // the path from sample i - L to sample i + L, summed in index order
                for (int q = -TRS_HILL_SPAN; q < TRS_HILL_SPAN; ++q) run += step[wrap(i + q)];
                const double rise = T.py[wrap(i + TRS_HILL_SPAN)] - T.py[wrap(i - TRS_HILL_SPAN)];
                slope[i] = std::atan(run > 1e-9 ? rise / run : 0.0);
            }
            for (int i = 0; i < n; ++i)
                T.dpitch[i] = (float)std::min(std::max(slope[wrap(i + TRS_HILL_AHEAD)] - slope[i], -(double)TRS_HILL_MAX_DPITCH), (double)TRS_HILL_MAX_DPITCH);
        }
        T.sky.assign((size_t)H, 0u);
        for (int v = 0; v < H; ++v) {
            const double g = std::min(((double)v + 0.5) / half_h, 1.0);
            uint32_t rgbv = 0;
            for (int ch = 0; ch < 3; ++ch) rgbv |= (uint32_t)round_colour((double)sky_top[ch] + ((double)sky_hor[ch] - (double)sky_top[ch]) * g) << (8 * ch);
            T.sky[v] = rgbv;
        }
        T.far_rgb = 0;
        for (int ch = 0; ch < 3; ++ch) T.far_rgb |= (uint32_t)round_colour((double)base[0][ch] * (1.0 - TRS_FOG_MAX) + (double)fog[ch] * TRS_FOG_MAX) << (8 * ch);
        T.inv_f = (float)(1.0 / f); T.hh = (float)half_h; T.pitch_f = (float)pitch;
        T.cam_h_f = (float)cfg.cam_h; T.z_far_f = (float)cfg.z_far; T.inv_zfar_f = (float)(1.0 / cfg.z_far); T.fog_f = (float)TRS_FOG_MAX;
    }
    return TRS_OK;
}

}  // namespace trsim
