#!/bin/bash
# round 4: where the dynamic-brightness step (26 us against 9.4 raw) spends its time; timing-only builds of raster_dyn_batch
cd "$(dirname "$0")/.."
echo "== product"; python scripts/filter_bench.py resident 2>/dev/null | grep -E "raw frames \(|dynamic"
for a in 1 2 3; do echo "== TRS_DYN_ABLATE=$a"; TRS_HIP_LIB=$PWD/scripts/ab_bin/libtrsim_dyn$a.so python scripts/filter_bench.py resident 2>/dev/null | grep -E "dynamic"; done
