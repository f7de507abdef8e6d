// Test driver (CPU): runs the PRODUCT's host-side table builder (triton-racer-sim_amd/csrc/trsim_tables.cpp) on a config and a
// track handed over in files, and writes the tables out; tests/test_host_tables.py builds it with AddressSanitizer +
// UBSan and compares the tables with the oracle's bit for bit.
//   host_tables_driver <config.bin> <points.bin> <out_prefix>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../triton-racer-sim_amd/csrc/trsim_tables.hpp"

template <typename T>
static void dump(const std::string& path, const std::vector<T>& v)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { std::perror(path.c_str()); std::exit(3); }
    std::fwrite(v.data(), sizeof(T), v.size(), f);
    std::fclose(f);
}

int main(int argc, char** argv)
{
    if (argc != 4) return 2;
    trs_config cfg;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(&cfg, sizeof cfg, 1, f) != 1) return 3;
    std::fclose(f);
    f = std::fopen(argv[2], "rb");
    if (!f) return 3;
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<double> xyz(bytes / sizeof(double));
    if (std::fread(xyz.data(), sizeof(double), xyz.size(), f) != xyz.size()) return 3;
    std::fclose(f);
    trsim::TrackTables t;
    std::string err;
    const int rc = trsim::build_tables(cfg, xyz.data(), (int)(xyz.size() / 3), t, err);
    if (rc) { std::fprintf(stderr, "build_tables: %d %s\n", rc, err.c_str()); return rc == 0 ? 1 : 10; }
    const std::string p = argv[3];
    dump(p + ".map", t.map); dump(p + ".rowtab", t.rowtab); dump(p + ".palette", t.palette); dump(p + ".tangent", t.tangent);
    dump(p + ".rowdepth", t.rowdepth);
    std::printf("%d %d %d %d\n", t.n_points, t.info.map_w, t.info.map_h, t.info.map_words);
    return 0;
}
