#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; collected separately, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) into profiles/pmc_traffic.json.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu
    python tools/pmc_traffic.py OUT profiles/pmc_traffic.json [fused_steps]

Units and gfx950 corrections (guide, section HBM): both counters are in KiB; FETCH_SIZE counts 64 B per
128-B request of a wide coalesced read, i.e. reports HALF the bytes -> doubled here (upper bound for
narrow reads); WRITE_SIZE is exact for 16-B-per-lane streaming stores.
"""
import csv
import glob
import json
import sys


def per_dispatch(root, counter, kernel_substr):
    vals = []
    for f in glob.glob(root + '/**/*_counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter and kernel_substr in r['Kernel_Name']:
                vals.append(float(r['Counter_Value']))
    return vals


def main():
    root, out = sys.argv[1], sys.argv[2]
    fused = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    kern = 'rollout'
    fetch = per_dispatch(root + '/fetch', 'FETCH_SIZE', kern)
    write = per_dispatch(root + '/write', 'WRITE_SIZE', kern)
    assert fetch and write, 'no rollout_kernel dispatches found'
    f = sum(fetch) / len(fetch) * 1024.0
    w = sum(write) / len(write) * 1024.0
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    from student_mechanism_design_amd import _lib
    rec = {
        'build_id': _lib.build_id(),      # digest of the sources of the library the counters were collected from
        'config': {'nodes': 200, 'police': 4, 'envs': 4096, 'fused': fused},
        'kernel': kern, 'dispatches': len(fetch),
        'fetch_size_bytes_raw': f, 'write_size_bytes': w,
        'fetch_bytes_corrected_x2': 2.0 * f,
        'hbm_bytes_per_launch': 2.0 * f + w,
        'note': 'FETCH_SIZE doubled per the gfx950 correction (exact only for wide coalesced reads: upper bound here); '
                'WRITE_SIZE exact for 16-B/lane stores',
    }
    json.dump(rec, open(out, 'w'), indent=1)
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
