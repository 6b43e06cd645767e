"""Per-kernel statistics of a rocprofv3 --kernel-trace CSV over the steady state: the first `skip_frac`
of every kernel's launches (warm-up steps: code-object loading, cold caches) is left out.

    python scripts/trace_stats.py <kernel_trace.csv> [skip_frac=0.3] > profiles/..._steady_stats.csv
"""
import csv
import sys


def main(path, skip_frac=0.3):
    per = {}
    for r in csv.DictReader(open(path)):
        per.setdefault(r['Kernel_Name'], []).append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
    out = []
    for name, spans in per.items():
        spans.sort()
        keep = spans[int(len(spans) * skip_frac):] or spans
        durs = [e - s for s, e in keep]
        out.append((sum(durs), name, len(durs), sum(durs) / len(durs), min(durs), max(durs)))
    total = sum(o[0] for o in out)
    w = csv.writer(sys.stdout)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
    for tot, name, n, avg, lo, hi in sorted(out, reverse=True):
        w.writerow([name, n, tot, '%.1f' % avg, '%.2f' % (100.0 * tot / total), lo, hi])


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.3)
