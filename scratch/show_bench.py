import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    r = d["roofline"]
    print(f.split("/")[-1], "value %.3fM" % (d["value"] / 1e6), "full", d.get("value_full_schedule"), "compact", d.get("value_compact_300K"), "kernel_ms %.4f" % r["kernel_ms"],
          "valu", (r.get("valu") or {}).get("busy_frac"), (r.get("valu") or {}).get("busy_frac_at_measured_clock"),
          "valu_compact", (r.get("valu_compact_300K") or {}).get("busy_frac"), (r.get("valu_compact_300K") or {}).get("busy_frac_at_measured_clock"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
