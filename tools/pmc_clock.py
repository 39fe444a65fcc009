#!/usr/bin/env python3
"""Clock the chip held per kernel symbol: GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration (MI355X_MICROARCH.md, "DVFS give-back").

    rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d DIR -o clk -- python3 tools/bf16_steps.py
    tools/pmc_clock.py DIR/clk_counter_collection.csv

Only dispatches of at least 0.3 ms count (the quotient reads high on shorter ones)."""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    try:
        dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    except (KeyError, ValueError):
        print("no Start_Timestamp / End_Timestamp columns in this rocprofv3 output:", list(r.keys()))
        sys.exit(1)
    if dur < 3e5:
        continue
    k = (r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hipvoc::", ""), r["Grid_Size"])
    a = acc[k]
    a[0] += float(r["Counter_Value"]); a[1] += dur; a[2] += 1
print("kernel, grid, dispatches, mean ms, GHz held")
for k, (cyc, ns, n) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]}, {k[1]}, {n}, {ns / n / 1e6:.3f}, {cyc / 8.0 / ns:.3f}")
