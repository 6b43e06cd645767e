"""Summarise rocprofv3 --pmc counter CSVs (one pass per counter) into per-kernel HBM bytes per
launch, applying the corrections of /opt/skills/guides/MI355X_MICROARCH.md (section HBM):
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-byte requests at 64 bytes for
wide coalesced streaming reads, so it is doubled (an UPPER bound for kernels whose reads are
narrower; stated in the output).

    python scripts/pmc_summary.py <dir with *counter_collection.csv> [more dirs] > profiles/...json
"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r'^void ', '', name)
    return re.sub(r'\(.*$', '', name)


def main(dirs):
    acc = {}
    for d in dirs:
        for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    k = short(row['Kernel_Name'])
                    c = row['Counter_Name']
                    e = acc.setdefault(k, {}).setdefault(c, [0.0, set()])
                    e[0] += float(row['Counter_Value'])
                    e[1].add((path, row['Dispatch_Id']))
    out = {}
    for k, counters in acc.items():
        rec = {}
        for c, (total, disp) in counters.items():
            rec[c + '_KiB_per_launch'] = total / max(len(disp), 1)
            rec[c + '_launches'] = len(disp)
        if 'FETCH_SIZE_KiB_per_launch' in rec and 'WRITE_SIZE_KiB_per_launch' in rec:
            rec['hbm_bytes_per_launch_raw'] = (rec['FETCH_SIZE_KiB_per_launch'] + rec['WRITE_SIZE_KiB_per_launch']) * 1024
            rec['hbm_bytes_per_launch_fetch_x2'] = (2 * rec['FETCH_SIZE_KiB_per_launch'] + rec['WRITE_SIZE_KiB_per_launch']) * 1024
        out[k] = rec
    # the whole step: every kernel's raw FETCH + WRITE bytes, per forward pass (one pixel_norm launch per forward)
    forwards = out.get('pixel_norm_kernel', {}).get('FETCH_SIZE_launches', 0)
    if forwards:
        total = sum(r.get('hbm_bytes_per_launch_raw', 0.0) * r.get('FETCH_SIZE_launches', 0) for r in out.values())
        out['__step__'] = dict(forwards=forwards, hbm_bytes_per_step_raw=total / forwards)
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == '__main__':
    main(sys.argv[1:])
