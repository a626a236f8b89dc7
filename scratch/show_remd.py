import json, sys
d = json.loads(sys.stdin.readline())
print('%s  %.3f M  remd_exchange_us %.1f  (iteration minus step kernel %.1f)' % (sys.argv[1], d["value"] / 1e6, d["remd_exchange_us"], d["remd_iteration_minus_step_kernel_us"]))
