set -e
mkdir -p gpurun_out/r3o
export BGR_LIB_PATH=$PWD/bgreat_amd/lib_phase/libbgreat_gpu.so
B="python bench.py --no-pmc --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 --steps 20 --warmup 3 --reads-per-step 131072"
for stop in 0 1 2; do for lds in 0 1; do
$B --debug-stop $stop --lds-mphf $lds > gpurun_out/r3o/s${stop}_l$lds.json 2> gpurun_out/r3o/s${stop}_l$lds.err
done; done
$B --waves 8 > gpurun_out/r3o/w8.json 2> gpurun_out/r3o/w8.err
$B --waves 4 > gpurun_out/r3o/w4.json 2> gpurun_out/r3o/w4.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3o/*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], [(k["kernel"][:24], k["avg_ms"]) for k in d["roofline"]["kernels_ms"]], d["config"]["launch"])
PY
