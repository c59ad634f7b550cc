#!/usr/bin/env python3
"""Static instruction mix of kernels in a hipcc -S listing.  usage: isa_mix.py file.s substring [substring...]"""
import re, sys, collections
txt = open(sys.argv[1]).read()
for m in re.finditer(r'\n(_Z\w+): +; @', txt):
    name = m.group(1)
    if not all(s in name for s in sys.argv[2:]):
        continue
    body = txt[m.end():txt.index('.Lfunc_end', m.end())]
    c = collections.Counter(mm.group(1) for mm in re.finditer(r'\n\s+([a-z][a-z_0-9]+)[ \t]', body))
    tot = sum(c.values())
    valu = sum(v for k, v in c.items() if k.startswith('v_') and 'mfma' not in k)
    print("%s\n  total %d  valu %d  salu %d  mfma %d  ds %d  vmem %d" % (name, tot, valu, sum(v for k, v in c.items() if k.startswith('s_')),
          sum(v for k, v in c.items() if 'mfma' in k), sum(v for k, v in c.items() if k.startswith('ds_')),
          sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'flat_', 'scratch_')))))
    print("  " + " ".join("%s:%d" % kv for kv in sorted(c.items(), key=lambda kv: -kv[1])[:40]))
