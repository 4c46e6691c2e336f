#!/bin/bash
# tools/ab_libs.sh LIB_A LIB_B [rounds] -- GPU box: the device-resident leg of the greedy workloads with two builds of the library in turn (same box, interleaved)
A=$1; B=$2; N=${3:-3}
for i in $(seq 1 $N); do
  for L in "$A" "$B"; do
    for w in ecoli small chr1; do
      v=$(BGR_LIB_PATH=$PWD/$L python bench.py --workload $w --no-sub --no-pmc --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['parity_sample']['gpu_equals_oracle'])")
      echo "$L $w $v"
    done
  done
done
