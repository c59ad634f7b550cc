#!/usr/bin/env python3
"""Global loads that are waited for (vmcnt(0)) before the next one is issued, per kernel of a hipcc -S listing: each is a full
memory round trip the wave sits out alone.  usage: isa_serial_loads.py file.s [substring]"""
import re, sys
txt = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'\n(_Z\w+): +; @', txt):
    if sub not in m.group(1):
        continue
    body = [l.strip() for l in txt[m.end():txt.index('.Lfunc_end', m.end())].split('\n') if l.strip() and not l.strip().startswith(';')]
    n = 0
    for i, l in enumerate(body):
        if l.startswith(('global_load_dword', 'buffer_load_dword')) and 'lds' not in l:
            nxt = body[i + 1:i + 4]
            if any(re.search(r's_waitcnt vmcnt\(0\)', x) for x in nxt) and not any(x.startswith(('global_load_dword', 'buffer_load_dword')) for x in nxt):
                n += 1
    short = re.sub(r'.*k_conv3x3I', 'conv<', m.group(1)).replace('EEEv11MpfConvArgs', '>').replace('ELi', ',').replace('Li', '')
    print("%-40s isolated load+wait pairs: %d" % (short[:40], n))
