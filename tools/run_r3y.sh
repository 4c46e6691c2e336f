set -e
mkdir -p gpurun_out/r3y
B="python bench.py --no-pmc --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 --steps 4 --warmup 1 --exhaustive --exh-first-pass-off"
for lib in lib lib_o5; do
  export BGR_LIB_PATH=$PWD/bgreat_amd/$lib/libbgreat_gpu.so
  $B --exh-search 1 --reads-per-step 1000000 > gpurun_out/r3y/${lib}_dfs_ecoli.json 2> gpurun_out/r3y/${lib}_dfs_ecoli.err
  $B --exh-search 1 --workload branchy --reads-per-step 200000 > gpurun_out/r3y/${lib}_dfs_branchy.json 2> gpurun_out/r3y/${lib}_dfs_branchy.err
  $B --exh-search 1 --exh-frame-cap 2 --workload branchy --reads-per-step 200000 > gpurun_out/r3y/${lib}_deep_branchy.json 2> gpurun_out/r3y/${lib}_deep_branchy.err
  $B --exh-search 2 --workload branchy --reads-per-step 500000 > gpurun_out/r3y/${lib}_dp_branchy.json 2> gpurun_out/r3y/${lib}_dp_branchy.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3y/*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], [(k["kernel"][:40], k["avg_ms"]) for k in d["roofline"]["kernels_ms"]], d["config"]["pass_counts_last_launch"], d["parity_sample"]["gpu_equals_oracle"])
PY
