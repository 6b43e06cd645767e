"""Print the kernels of a rocprofv3 --kernel-trace CSV in launch order with durations (us)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for r in rows[skip:]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    print('%10.1f  %s  grid=%s wg=%s' % (d, r['Kernel_Name'][:80], r.get('Grid_Size_X', r.get('Grid_Size', '?')),
                                         r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))))
