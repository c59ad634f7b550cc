#!/bin/bash
# Per-kernel register / occupancy summary (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
# usage: ./resource_usage.sh mpf_render.hip
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FLAGS -Rpass-analysis=kernel-resource-usage -c "$1" -o /dev/null 2>&1 | python3 -c '
import sys,re,subprocess
cur={}
for l in sys.stdin:
    m=re.search(r"remark: (?:[^:]+:\d+:\d+: )?\s*([A-Za-z ]+): (\S+)", l)
    if not m: continue
    k,v=m.group(1).strip(),m.group(2)
    if k=="Function Name":
        if cur: print(cur)
        name=subprocess.run(["c++filt",v],capture_output=True,text=True).stdout.strip().split("(")[0]
        cur={"kernel":name}
    elif k in ("VGPRs","AGPRs","TotalSGPRs","ScratchSize [bytes/lane]","Occupancy [waves/SIMD]","LDS Size [bytes/block]","ScratchSize","Occupancy","LDS Size"): cur[k]=v
if cur: print(cur)
'
