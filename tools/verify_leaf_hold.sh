set -u
O=gpurun_out/r4q; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
grep -q " passed" $O/gpu_tests.log && ! grep -qi "failed\|fault" $O/gpu_tests.log || exit 1
tools/leaf_hold_sweep.sh "0" > $O/leaf_hold_default.txt 2>&1; cat $O/leaf_hold_default.txt
