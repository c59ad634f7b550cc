#!/usr/bin/env python3
"""Registers / scratch / LDS per kernel from a hipcc -S listing.  usage: isa_regs.py file.s [substring]"""
import re, sys
txt = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
rows = {}
for m in re.finditer(r'\.set (_Z\w+)\.(num_vgpr|num_agpr|private_seg_size), (\d+)', txt):
    rows.setdefault(m.group(1), {})[m.group(2)] = int(m.group(3))
for k, v in rows.items():
    if sub in k:
        short = re.sub(r'.*k_conv3x3I', 'conv<', k).replace('EEEv11MpfConvArgs', '>').replace('ELi', ',').replace('Li', '')
        print("%-44s vgpr %3d agpr %3d scratch %4d" % (short[:44], v.get('num_vgpr', -1), v.get('num_agpr', -1), v.get('private_seg_size', -1)))
