set -e
mkdir -p gpurun_out/r3s
BGREAT_TIMING=1 python tools/e2e.py --reads 100000000 --check 0 --batch 262144 > gpurun_out/r3s/e2e.json 2> gpurun_out/r3s/e2e.err || { tail -20 gpurun_out/r3s/e2e.err; exit 1; }
grep "bgreat:" gpurun_out/r3s/e2e.err | head -14; cat gpurun_out/r3s/e2e.json
