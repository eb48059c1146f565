#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs per kernel (per dispatch and per env-step)."""
import collections
import csv
import glob
import sys


def main(root, env_steps_per_dispatch):
    for f in sorted(glob.glob(root + '/pmc_*/*/*_counter_collection.csv')):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in rows:
            k = r['Kernel_Name'][:48]
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            disp[k].add(r['Dispatch_Id'])
        for k, v in agg.items():
            if 'rollout' in k or 'engine' in k or 'step_kernel' in k:
                n = len(disp[k])
                print(f.split('/')[-3], k, 'dispatches', n)
                for c, val in sorted(v.items()):
                    print(f'   {c:28s} {val / n:16.0f} /dispatch {val / n / env_steps_per_dispatch:10.1f} /env-step')


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 4096 * 64)
