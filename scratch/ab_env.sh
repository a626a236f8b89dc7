# A/B of one environment knob in the product library: bash scratch/ab_env.sh VAR [bench args]   (values 1 and 0, two rounds, extended and compact)
set -o pipefail
var=$1; shift
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for k in 1 0 1 0; do for leg in "" "--folded"; do
env $var=$k timeout -k 10 300 python bench.py --no-cpu-baseline --no-schedule-legs --steps 150 --warmup 5 $leg "$@" > /tmp/b.json 2>/tmp/b.err || { cat /tmp/b.err; exit 2; }
python -c "import json; d=json.load(open('/tmp/b.json')); print('[$var=$k ${leg:-extended}] value %.3fM  kernel_ms %.4f' % (d['value']/1e6, d['roofline']['kernel_ms']))"
done; done
