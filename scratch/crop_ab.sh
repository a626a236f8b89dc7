set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cropped or golden" --timeout 200 --timeout-method thread 2>&1 | tail -5 &&
for w in 2xov 1r69 t0958; do for c in 1 0; do
 AWSEM_INFO=1 AWSEM_FM_CROP=$c timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps 40 --warmup 5 --no-schedule-legs 2>$O/crop_${w}_$c.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$w crop=$c  %.3f M  kernel_ms %.4f' % (d['value']/1e6, d['roofline']['kernel_ms']))" || exit 1
 grep "fragment memory" $O/crop_${w}_$c.err | head -1
done; done
