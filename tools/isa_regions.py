#!/usr/bin/env python3
"""Instruction mix of a kernel's prologue / loops / epilogue from a hipcc -S listing.  usage: isa_regions.py file.s mangled-name-substring"""
import re, sys, collections
txt = open(sys.argv[1]).read()
m = [mm for mm in re.finditer(r'\n(_Z\w+): +; @', txt) if sys.argv[2] in mm.group(1)][0]
body = txt[m.end():txt.index('.Lfunc_end', m.end())].split('\n')
labels = {}
for k, l in enumerate(body):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        labels[mm.group(1)] = k
loops = []
for k, l in enumerate(body):
    mm = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l) or re.search(r's_branch (\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
        loops.append((labels[mm.group(1)], k))


def mix(a, b, tag):
    c = collections.Counter()
    for l in body[a:b]:
        l = l.strip()
        if l and l[0] not in ';.':
            c[l.split()[0]] += 1
    print("%s lines %d-%d: valu %d mfma %d ds %d vmem %d salu %d" % (tag, a, b, sum(v for k, v in c.items() if k.startswith('v_') and 'mfma' not in k),
          sum(v for k, v in c.items() if 'mfma' in k), sum(v for k, v in c.items() if k.startswith('ds_')),
          sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'flat_', 'scratch_'))), sum(v for k, v in c.items() if k.startswith('s_'))))
    print("   " + " ".join("%s:%d" % kv for kv in c.most_common(40)))


print(m.group(1), "loops", loops)
mix(0, loops[0][0] if loops else len(body), "prologue")
for a, b in loops:
    mix(a, b, "loop")
if loops:
    mix(max(b for _, b in loops), len(body), "epilogue")
