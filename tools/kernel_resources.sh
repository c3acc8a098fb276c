#!/bin/bash
# Prints VGPR/SGPR/LDS/occupancy per kernel of the product library (hipcc -Rpass-analysis).
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c radix_sorting_amd/csrc/rsx.hip -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re,sys,subprocess
rows=[];cur={}
for line in sys.stdin:
    m=re.search(r"remark: +(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)",line)
    if not m: continue
    k,v=m.groups()
    if k=="Function Name":
        cur={"name":v}; rows.append(cur)
    else: cur[k.split()[0]]=v
for r in rows:
    name=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip()
    name=re.sub(r"\(.*","",name).replace("rsx::","").replace("unsigned long long","u64").replace("unsigned int","u32").replace("unsigned short","u16").replace("unsigned char","u8")
    print("%-58s vgpr %4s sgpr %4s scratch %4s lds %6s occ %s"%(name[:58],r.get("VGPRs"),r.get("TotalSGPRs"),r.get("ScratchSize"),r.get("LDS"),r.get("Occupancy")))
'
