#!/usr/bin/env python3
"""What the amdsmi binding reports on this box for GPU 0 (violation accumulators, gpu_metrics): the raw material of bench.ThrottleProbe."""
import json
import sys

try:
    import amdsmi

    amdsmi.amdsmi_init()
    for h in amdsmi.amdsmi_get_processor_handles():
        out = {"bdf": amdsmi.amdsmi_get_gpu_device_bdf(h)}
        for fn in ("amdsmi_get_violation_status", "amdsmi_get_gpu_metrics_info", "amdsmi_get_power_cap_info"):
            try:
                out[fn] = getattr(amdsmi, fn)(h)
            except Exception as e:  # noqa: BLE001
                out[fn] = f"{type(e).__name__}: {e}"
        print(json.dumps(out, default=str)[:6000])
except Exception as e:  # noqa: BLE001
    print("amdsmi unavailable:", type(e).__name__, e)
    sys.exit(0)
