# fp64 check mode: product against variants (GPU box).  Usage: bash scratch/ab_fp64.sh <name> ...
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for name in "" "$@"; do
  lib=""; [ -n "$name" ] && lib=$R/.variants/libawsem_$name.so
  AWSEM_HIP_LIBRARY=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-schedule-legs --precision fp64 --steps 20 --warmup 3 > /tmp/b.json 2> /tmp/b.err || { cat /tmp/b.err; exit 2; }
  python -c "import json; d=json.load(open('/tmp/b.json')); print('[%-10s fp64] value %.3fM steps/s  kernel_ms %.4f' % ('${name:-product}', d['value']/1e6, d['roofline']['kernel_ms']))"
done
