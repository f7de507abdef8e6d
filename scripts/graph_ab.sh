set -e
export TMPDIR=/tmp
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; }
echo "graph  1024:"; run --envs-per-gpu 1024 --steps 2000
echo "eager  1024:"; TRS_NO_GRAPH=1 run --envs-per-gpu 1024 --steps 2000
echo "eager  4096:"; TRS_NO_GRAPH=1 run --envs-per-gpu 4096 --steps 500
mkdir -p gpurun_out/tl_graph gpurun_out/tl_eager
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_graph -o t -- python3 bench.py --no-cpu-baseline --steps 200 --warmup 40 > /dev/null 2>&1 || true
TRS_NO_GRAPH=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_eager -o t -- python3 bench.py --no-cpu-baseline --steps 200 --warmup 40 > /dev/null 2>&1 || true
echo "--- graph timeline"; python3 scripts/timeline.py gpurun_out/tl_graph
echo "--- eager timeline"; python3 scripts/timeline.py gpurun_out/tl_eager
