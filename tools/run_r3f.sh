set -e
mkdir -p gpurun_out/r3f
python3 tools/text_bench.py 524288 4 > gpurun_out/r3f/tb.txt 2>&1 || { tail -20 gpurun_out/r3f/tb.txt; exit 1; }
grep rep gpurun_out/r3f/tb.txt
for b in 0 262144 1048576; do
BGREAT_TIMING=1 python tools/e2e.py --reads 100000000 --check 0 --batch $b > gpurun_out/r3f/e2e_b$b.json 2> gpurun_out/r3f/e2e_b$b.err || { tail -20 gpurun_out/r3f/e2e_b$b.err; exit 1; }
grep "stage busy\|pool CPU" gpurun_out/r3f/e2e_b$b.err | tail -2; cat gpurun_out/r3f/e2e_b$b.json
done
