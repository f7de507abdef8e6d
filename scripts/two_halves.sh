set -e
mkdir -p gpurun_out
python3 scripts/two_halves.py 120 160 0 1024 > gpurun_out/two_halves.txt 2>&1
python3 scripts/two_halves.py 240 320 1 512 >> gpurun_out/two_halves.txt 2>&1
cat gpurun_out/two_halves.txt
