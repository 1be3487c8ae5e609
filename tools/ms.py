import json, sys
label = sys.argv[1] if len(sys.argv) > 1 else ""
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        r = d.get("roofline") or {}
        print(label, "ms/step", d.get("ms_per_step"), "median", d.get("ms_per_step_median_events"),
              "| dom", r.get("kernel"), r.get("avg_us"), "us frac", r.get("frac"),
              "| loss", (d.get("config") or {}).get("final_l1_loss", (d.get("config") or {}).get("final_loss_gen")))
        kb = d.get("kernel_breakdown_ms")
        if kb:
            for k, v in kb.items():
                print("    ", k, v)
