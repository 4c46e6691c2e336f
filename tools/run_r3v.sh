set -e
bash tools/prof.sh r03_ecoli --e2e-reads 0 --pcie-steps 0 > gpurun_out/prof_r03_ecoli.log 2>&1 || { tail -20 gpurun_out/prof_r03_ecoli.log; exit 1; }
tail -60 gpurun_out/prof_r03_ecoli/summary.txt
