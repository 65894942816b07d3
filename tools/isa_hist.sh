#!/bin/bash
# Compiles dr_mvsnet.hip with -save-temps into build/tmp and prints an instruction histogram of one kernel.
# Usage: tools/isa_hist.sh <mangled-name-substring> [ops...]
cd /root/repo && mkdir -p build/tmp && cd build/tmp || exit 1
if [ -z "$NO_BUILD" ]; then
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-function -Wno-pass-failed -save-temps -c ../../tandem_amd/csrc/dr_mvsnet.hip -o m2.o 2>&1 | grep -E "error|warning: v"
fi
python3 - "$@" <<'PY'
import re, sys
from collections import Counter
s = open('dr_mvsnet-hip-amdgcn-amd-amdhsa-gfx950.s').read()
names = [n for n in re.findall(r'^(_Z\w+):', s, re.M) if sys.argv[1] in n]
for name in names:
    i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
    body = s[i:j]
    open(name[:60] + '.s', 'w').write(body)
    c = Counter(re.findall(r'^\s+([a-z_0-9]+)', body, re.M))
    m = re.search(r'\.vgpr_count:\s+(\d+)', s[s.index('.name:           ' + name):][:3000]) if ('.name:           ' + name) in s else None
    print(name, 'instrs', sum(c.values()))
    keys = sys.argv[2:] or [k for k, _ in c.most_common(40)]
    print('  ' + '  '.join('%s=%d' % (k, c.get(k, 0)) for k in keys))
PY
