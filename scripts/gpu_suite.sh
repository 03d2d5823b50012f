#!/bin/bash
# One GPU-box call: the -m gpu tests, then (only if they pass) any extra test files and the clock-independent tick count.
#   scripts/gpu_suite.sh <tag> [extra pytest files...]
# Every step has its own time limit; nothing more is started on the GPU after a failure.
tag=$1; shift
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > gpurun_out/${tag}_gpu_tests.log 2>&1
rc=$?
tail -4 gpurun_out/${tag}_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
for f in "$@"; do
    timeout -k 10 200 python -u -m pytest "$f" -x -q --timeout 150 --timeout-method thread > gpurun_out/${tag}_$(basename $f .py).log 2>&1
    rc=$?
    tail -4 gpurun_out/${tag}_$(basename $f .py).log
    [ $rc -eq 0 ] || exit $rc
done
if [ -f build_diag/librt_is_ticks.so ]; then
    RT_IS_LIB=build_diag/librt_is_ticks.so timeout -k 10 150 python scripts/ticks.py > gpurun_out/${tag}_ticks.txt 2>&1
    rc=$?
    cat gpurun_out/${tag}_ticks.txt
    [ $rc -eq 0 ] || exit $rc
fi
