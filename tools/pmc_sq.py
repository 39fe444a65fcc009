#!/usr/bin/env python3
"""Summarise one rocprofv3 --pmc pass of SQ counters per (kernel symbol, grid size).

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \
              SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d DIR -- python3 tools/profile_steps.py
    tools/pmc_sq.py DIR/*_counter_collection.csv

mfma_busy/busy: SQ_VALU_MFMA_BUSY_CYCLES is summed over SIMDs, SQ_BUSY_CYCLES over shader engines: the ratio saturates
at 32 (all four SIMDs of every CU busy all the time), so mfma_util = ratio / 32.
"""
import collections
import csv
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hipvoc::", ""), r["Grid_Size"])
    rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        disp[k] += 1
print("kernel, grid, dispatches, wait_any, wait_inst, active, wait_lds, mfma_util, bank_conflict/wave_cycles")
for k, c in sorted(rows.items(), key=lambda kv: -kv[1]["SQ_BUSY_CYCLES"]):
    wc = max(c["SQ_WAVE_CYCLES"], 1.0)
    print(f"{k[0]}, {k[1]}, {disp[k]}, {c['SQ_WAIT_ANY']/wc:.2f}, {c['SQ_WAIT_INST_ANY']/wc:.2f}, {c['SQ_ACTIVE_INST_ANY']/wc:.2f}, "
          f"{c['SQ_WAIT_INST_LDS']/wc:.3f}, {c['SQ_VALU_MFMA_BUSY_CYCLES']/max(c['SQ_BUSY_CYCLES'],1)/32:.3f}, "
          f"{c['SQ_LDS_BANK_CONFLICT']/wc:.4f}")
