#!/bin/bash
# tools/ab_exh.sh LIB_A LIB_B [rounds] -- GPU box: the exhaustive workloads with two builds of the library in turn (same box, interleaved)
A=$1; B=$2; N=${3:-2}
for i in $(seq 1 $N); do
  for L in "$A" "$B"; do
    for w in "--workload branchy" "--exhaustive" "--workload chr1 --exhaustive"; do
      v=$(BGR_LIB_PATH=$PWD/$L python bench.py $w --no-sub --no-pmc --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 --cpu-sample-exh 0 --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['parity_sample']['gpu_equals_oracle'])")
      echo "$L [$w] $v"
    done
  done
done
