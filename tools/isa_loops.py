#!/usr/bin/env python3
"""Per-kernel summary of a hipcc -S --cuda-device-only listing: total VALU / scratch ops and, for every loop with > 50 VALU ops, its
instruction mix (VALU, buffer / global loads, scalar loads, stores, s_waitcnt).  usage: isa_loops.py listing.s <mangled-name-prefix> ..."""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")


def func(start_pat):
    i = [k for k, l in enumerate(lines) if l.startswith(start_pat)][0]
    j = i
    while not lines[j].startswith(".Lfunc_end"):
        j += 1
    return lines[i:j]


def summarize(name, f):
    nv = sum(1 for l in f if re.match(r"\s+v_", l))
    scr = sum(1 for l in f if "scratch_" in l)
    print(name, "lines", len(f), "valu", nv, "scratch", scr)
    labels = {}
    for k, l in enumerate(f):
        m = re.match(r"(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = k
    for k, l in enumerate(f):
        m = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < k:
            body = f[labels[m.group(1)]:k]
            c = lambda pat: sum(1 for x in body if re.match(pat, x))   # noqa: E731
            v = c(r"\s+v_")
            if v > 50 and len(body) < 1500:
                print("   loop", m.group(1), "len", len(body), "valu", v, "bufload4", c(r"\s+buffer_load_dwordx4"), "waitcnt", c(r"\s+s_waitcnt"),
                      "s_load", c(r"\s+s_load"), "global_load", c(r"\s+global_load"), "store", c(r"\s+global_store"))


for pat in sys.argv[2:]:
    summarize(pat, func(pat))
