set -e
mkdir -p gpurun_out/r3x
BGR_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --reads-per-step 300000 --e2e-reads 600000 --no-pmc --cpu-sample 0 --pcie-steps 0 > gpurun_out/r3x/rehearse2.json 2> gpurun_out/r3x/rehearse2.err || { tail -30 gpurun_out/r3x/rehearse2.err; exit 1; }
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r3x/rehearse2.json") if l.strip().startswith("{")][-1])
print(d["value"], d["n_gpus"], d["e2e"])
PY
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3x/pytest.txt 2>&1 || { tail -40 gpurun_out/r3x/pytest.txt; exit 1; }
tail -3 gpurun_out/r3x/pytest.txt
