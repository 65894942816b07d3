"""Merge TUNED rows that repeat across autotune logs into tandem_amd/csrc/conv_tuned.h: python tools/merge_tuned.py log1 log2 [...]
A row is taken when the same plan won for the same layer signature in at least two of the logs (tools/tune_conv.sh's rule)."""
import collections, re, sys
rows = collections.defaultdict(list)
for path in sys.argv[1:]:
    for line in open(path):
        if line.startswith('TUNED'):
            body = line[len('TUNED'):].rstrip()
            vals = [v.strip() for v in re.match(r'\s*\{([^}]*)\}', body).group(1).split(',')]
            rows[','.join(vals[:13])].append((','.join(vals[13:]), body))
keep = {}
for key, lst in rows.items():
    plan, n = collections.Counter(p for p, _ in lst).most_common(1)[0]
    if n >= 2:
        keep[key] = [b for p, b in lst if p == plan][-1]
p = 'tandem_amd/csrc/conv_tuned.h'
out, seen = [], set()
for line in open(p).read().split('\n'):
    m = re.match(r'\s*\{([^}]*)\},', line)
    if m:
        key = ','.join(v.strip() for v in m.group(1).split(',')[:13])
        if key in keep:
            if key not in seen:
                out.append(keep[key]); seen.add(key)
            continue
    out.append(line)
new = [keep[k] for k in keep if k not in seen]
idx = [i for i, l in enumerate(out) if 'sentinel' in l][0]
# new rows go to the end of their shape's section: 480x640 rows before the 320x512 header
sec = [i for i, l in enumerate(out) if '---- 320x512' in l][0]
head = [r for r in new if ', 480, 640,' in r or ', 240, 320,' in r or ', 120, 160,' in r or ', 60, 80,' in r or ', 30, 40,' in r or ', 15, 20,' in r]
rest = [r for r in new if r not in head]
out = out[:sec] + head + out[sec:idx] + rest + out[idx:]
open(p, 'w').write('\n'.join(out))
print(len(new), 'new,', len(seen), 'replaced')
