"""Instruction mix of the largest backward-branch loop of each kernel whose mangled name contains argv[1] (reads build/tmp/*.s written by tools/isa_hist.sh)."""
import re, sys, glob
from collections import Counter
for path in sorted(glob.glob('*.s')):
    if path.startswith('dr_') or not all(a in path for a in sys.argv[1:]): continue
    lines = open(path).read().split('\n')
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m: labels[m.group(1)] = i
    best = None
    for i, l in enumerate(lines):
        m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
        if m and labels.get(m.group(1), 1e9) < i:
            span = (labels[m.group(1)], i)
            if best is None or span[1] - span[0] > best[1] - best[0]: best = span
    if not best: continue
    body = [l.strip().split()[0] for l in lines[best[0]:best[1] + 1] if re.match(r'^\s+[a-z]', l)]
    c = Counter(body)
    g = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    print(path[:50], 'loop', best, 'instrs', len(body), 'VALU', g('v_'), 'SALU', g('s_'), 'VMEM', g('global_') + g('buffer_'), 'DS', g('ds_'))
    print('   ', ' '.join('%s=%d' % kv for kv in c.most_common(22)))
