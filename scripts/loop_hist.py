"""Opcode histogram of the innermost loops of one kernel in a hipcc -S listing: python scripts/loop_hist.py file.s kernel_substring [min_len]"""
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 100
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and name in l and l.rstrip().endswith(':') or (name in l and re.match(r'^_Z\S+:', l)))
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
labels = {}
for i in range(start, end):
    m = re.match(r'^(\.LBB\d+_\d+):', lines[i])
    if m: labels[m.group(1)] = i
for i in range(start, end):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', lines[i]) or re.search(r's_branch\s+(\.LBB\d+_\d+)', lines[i])
    if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] >= minlen:
        a = labels[m.group(1)]
        ops = collections.Counter()
        for l in lines[a:i + 1]:
            t = l.strip()
            if not t or t.startswith(';') or t.startswith('.'): continue
            ops[t.split()[0]] += 1
        print('loop %s: lines %d..%d (%d)' % (m.group(1), a - start, i - start, i - a))
        print('  ' + ', '.join('%s %d' % kv for kv in ops.most_common(18)))
