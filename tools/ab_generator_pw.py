#!/usr/bin/env python3
"""The generator record of bench.py (gen_3dphoto_dynamic.py end to end, 320 images) with one plane per workgroup in the fast engine against the walking table,
alternating on one box.  usage: python tools/ab_generator_pw.py [rounds]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ONE = "l7=1,l8s=1,l9=1,up0_0=1,up1_0=1,disp0=1"
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    for label, env in (("one plane per workgroup", ONE), ("walking table (default)", "")):
        os.environ["MPIFLOW_PW"] = env                     # inherited by the generator's process
        g = bench.generator_record()
        print("%-28s steady state %.1f pairs/s   whole process %.1f pairs/s   start-up %.2f s" % (label, g["pairs_per_s_steady_state"], g["pairs_per_s_whole_process"], g["startup_seconds"]), flush=True)
