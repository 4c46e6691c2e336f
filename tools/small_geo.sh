mkdir -p gpurun_out/r05h
for geo in "" "--waves 12" "--waves 8 --blocks-per-cu 3" "--waves 16 --blocks-per-cu 1" "--waves 8 --blocks-per-cu 2"; do
  for n in 262144 131072; do
    v=$(timeout -k 10 200 python bench.py --reads-per-step $n $geo --no-sub --no-pmc --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['parity_sample']['gpu_equals_oracle'])")
    echo "geo[$geo] n=$n: $v"
  done
done
