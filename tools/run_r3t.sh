set -e
mkdir -p gpurun_out/r3t
for x in 0 3 6; do
BGREAT_EXTRA_SETS=$x BGREAT_TIMING=1 python tools/e2e.py --reads 100000000 --check 0 --batch 262144 > gpurun_out/r3t/e2e_x$x.json 2> gpurun_out/r3t/e2e_x$x.err || { tail -20 gpurun_out/r3t/e2e_x$x.err; exit 1; }
echo "extra sets $x"; grep "bgreat: text calls\|stage busy" gpurun_out/r3t/e2e_x$x.err | head -3; python3 -c "
import json; d=json.load(open('gpurun_out/r3t/e2e_x$x.json')); print(d['run0']['mreads_per_s'], d['run1']['mreads_per_s'], d['run1'])"
done
