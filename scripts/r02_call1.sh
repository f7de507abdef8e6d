#!/bin/bash
# round 2, first GPU call: resident-mode parity, the whole GPU suite on the refactored kernels, bench A/B against the round-1 build
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r02_call1
mkdir -p $O
echo "== resident tests" && timeout -k 10 420 python -m pytest tests/test_resident.py -x -q > $O/resident_tests.log 2>&1; rc=$?; tail -15 $O/resident_tests.log; [ $rc -eq 0 ] || { echo "resident tests rc=$rc"; exit $rc; }
echo "== bench new build" && timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_new.json 2> $O/bench_new.err; rc=$?; cat $O/bench_new.json; [ $rc -eq 0 ] || { tail -5 $O/bench_new.err; exit $rc; }
echo "== bench new build, resident" && timeout -k 10 300 python bench.py --no-cpu-baseline --no-also --resident > $O/bench_resident.json 2> $O/bench_resident.err; rc=$?; cat $O/bench_resident.json; [ $rc -eq 0 ] || { tail -5 $O/bench_resident.err; exit $rc; }
echo "== bench 512 envs" && timeout -k 10 300 python bench.py --no-cpu-baseline --no-also --envs-per-gpu 512 > $O/bench_512.json 2>> $O/bench_new.err && cat $O/bench_512.json
echo "== bench 512 envs resident" && timeout -k 10 300 python bench.py --no-cpu-baseline --no-also --envs-per-gpu 512 --resident > $O/bench_512_res.json 2>> $O/bench_new.err && cat $O/bench_512_res.json
echo "== bench 20 steps (driver settings) launch / resident" && timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 5 > $O/bench_20.json 2>> $O/bench_new.err && cat $O/bench_20.json && timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 5 --resident > $O/bench_20_res.json 2>> $O/bench_new.err && cat $O/bench_20_res.json
echo "== full gpu suite" && timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; tail -8 $O/gpu_tests.log; exit $rc
