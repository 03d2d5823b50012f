#!/bin/bash
# One GPU-box call: one bench line per BASELINE row that is not the default (bench.py --workload / --g-units), each with
# its roofline and CPU legs.   scripts/gpu_workloads.sh <tag>
tag=$1
only=$2          # optional: run only the lines whose name contains this
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
run() {   # name, time limit, bench args...
    name=$1; lim=$2; shift; shift
    case "$name" in *"$only"*) ;; *) return 0;; esac
    timeout -k 10 $lim python3 $R/bench.py "$@" > $R/gpurun_out/${tag}_$name.json 2> $R/gpurun_out/${tag}_$name.err
    rc=$?
    [ $rc -eq 0 ] || { echo "$name failed ($rc)"; tail -5 $R/gpurun_out/${tag}_$name.err; exit $rc; }
    python3 -c "
import json,sys
d=json.load(open('$R/gpurun_out/${tag}_$name.json'))
print('$name', round(d['value']/1e6,2), 'M evals/s', round(d['roofline']['achieved'],1), 'TF', round(d['roofline']['frac'],4), d['roofline']['kernel'], 'cpu', round(d['cpu_baseline']['value']), d['result_check'])
"
}
run involute 300 --workload involute --steps 5 --warmup 2 || exit 1
run indep_gmm 400 --workload indep-gmm --steps 3 --warmup 1 || exit 1
for w in 64 128 256; do
    run hepmass_g$w 300 --workload hepmass --g-units $w --points 8192 --steps 3 --warmup 1 --cpu-points 32 --cpu-points-numpy 8 || exit 1
done
# width 1024 runs at 7.9 M sample-evals/s (4x the flops per eval): 2048 points x 40000 = 10 s per launch
run hepmass_g1024 400 --workload hepmass --g-units 1024 --points 2048 --steps 2 --warmup 1 --cpu-points 8 --cpu-points-numpy 2 || exit 1
run full_bank 900 --workload bank --steps 1 --warmup 1 --no-live-traffic --cpu-points 48 --cpu-points-numpy 16 || exit 1
run full_hepmass 900 --workload hepmass --steps 1 --warmup 1 --no-live-traffic --cpu-points 48 --cpu-points-numpy 16 || exit 1
# the per-GPU shares of the 8-GPU configurations at full size (5e9 sample-evals each: 19-20 bounded launches, ~160 s)
run full_arem 900 --workload arem --steps 1 --warmup 0 --no-live-traffic --cpu-points 48 --cpu-points-numpy 16 || exit 1
run full_casp 900 --workload casp --steps 1 --warmup 0 --no-live-traffic --cpu-points 48 --cpu-points-numpy 16 || exit 1
run full_ypmsd_weak 900 --workload ypmsd-weak --steps 1 --warmup 0 --no-live-traffic --cpu-points 48 --cpu-points-numpy 16 || exit 1
