#!/usr/bin/env python3
"""Per-env-step SQ counters of the rollout kernels from rocprofv3 counter CSVs (one or more files)."""
import collections
import csv
import sys

env_steps = float(sys.argv[1])
for f in sys.argv[2:]:
    agg = collections.defaultdict(float)
    disp = set()
    for r in csv.DictReader(open(f)):
        if 'rollout' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value'])
            disp.add(r['Dispatch_Id'])
    n = max(len(disp), 1)
    for c, v in sorted(agg.items()):
        print(f'{c:26s} {v / n:16.0f} /dispatch {v / n / env_steps:10.1f} /env-step')
