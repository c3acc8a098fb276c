#!/bin/bash
# kres.sh <file.hip> [name-filter]: VGPR/SGPR/scratch/LDS/occupancy of the kernels one translation unit instantiates
# (hipcc -Rpass-analysis=kernel-resource-usage; seconds for a probe that instantiates a few kernels, minutes for rsx.hip).
F=$1; PAT=${2:-.}
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iradix_sorting_amd/csrc -c "$F" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | PAT="$PAT" python3 -c '
import re,sys,subprocess,os
rows=[];cur={}
for line in sys.stdin:
    if " error" in line: print(line.rstrip())
    m=re.search(r"remark: +(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs Spill|VGPRs Spill): (\S+)",line)
    if not m: continue
    k,v=m.groups()
    if k=="Function Name":
        cur={"name":v}; rows.append(cur)
    else: cur[k.replace(" Spill","Spill").split()[0]]=v
pat=re.compile(os.environ["PAT"])
for r in rows:
    name=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip()
    name=re.sub(r"\(.*","",name).replace("rsx::","").replace("unsigned long long","u64").replace("unsigned int","u32").replace("unsigned short","u16").replace("unsigned char","u8").replace("void ","")
    if not pat.search(name): continue
    print("%-96s vgpr %4s sgpr %4s scratch %4s sspill %3s vspill %3s lds %6s occ %s"%(name[:96],r.get("VGPRs"),r.get("TotalSGPRs"),r.get("ScratchSize"),r.get("SGPRsSpill"),r.get("VGPRsSpill"),r.get("LDS"),r.get("Occupancy")))
'
