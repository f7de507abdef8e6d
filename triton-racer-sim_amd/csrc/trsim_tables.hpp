// trsim_tables.hpp — host-side (one-time, at trs_load_track) construction of the read-only tables the
// step kernel stages into LDS: per-point tangents / start headings, the packed 2-bit surface-class map,
// the per-image-row camera table and the per-row fogged palette.  Follows include/trsim_spec.h; all of
// it is binary64 arithmetic with IEEE +,-,*,/,sqrt,floor plus libm tan/sin/cos/atan2/fmod evaluated on
// the host, built with -ffp-contract=off.
//
// Track input: the raw [x,y,z] samples the reference's LocationTracker loads
// (TritonRacerSim/components/track_data_process.py:72-73), duplicates kept.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/trsim.h"

namespace trsim {

struct TrackTables {
    int n_points = 0;
    std::vector<double> px, py, pz;        // raw points (SoA, binary64)
    std::vector<float> tangent;            // [n][2] unit (tx, tz)
    std::vector<float> start_yaw;          // [n]
    trs_map_info info{};
    std::vector<uint32_t> map;             // [map_h][map_words], 16 cells per word, cell ix at bits 2*(ix&15)
    std::vector<float> rowtab;             // [H][2]
    std::vector<float> rowdepth;           // [H] z-depth of the ground plane per image row (z_far for sky / far rows)
    std::vector<uint32_t> palette;         // [H][4] 0x00BBGGRR
    float map_x0f = 0, map_z0f = 0, inv_cellf = 0;
    // tracks with elevation (include/trsim_spec.h): the per-point view-pitch offsets and what the kernels need to evaluate a frame's row tables per env
    bool hills = false;
    std::vector<float> dpitch;             // [n]; all zeros on a flat track
    std::vector<uint32_t> sky;             // [H] sky colour of every image row
    uint32_t far_rgb = 0;                  // colour of the rows beyond the far plane
    float inv_f = 0, hh = 0, pitch_f = 0, cam_h_f = 0, z_far_f = 0, inv_zfar_f = 0, fog_f = 0;
    // nearest-point accelerator (trsim_spec.h R3): points bucketed by (x, z) cell, ascending index inside a cell
    int grid_nx = 0, grid_nz = 0;
    double grid_x0 = 0, grid_z0 = 0;
    std::vector<uint16_t> grid_start;      // [grid_nx * grid_nz + 1]
    std::vector<uint16_t> grid_pts;        // [n_points]
};

// returns TRS_OK or a negative trs_status, message in `err`
int build_tables(const trs_config& cfg, const double* xyz, int n_points, TrackTables& out, std::string& err);

}  // namespace trsim
