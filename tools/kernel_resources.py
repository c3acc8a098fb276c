import sys,re,subprocess
out=subprocess.run(sys.argv[1:],capture_output=True,text=True).stderr
cur=None
rows=[]
for l in out.splitlines():
    m=re.search(r'Function Name: (\S+)',l) or re.search(r' Name: (\S+)',l)
    if m:
        cur={'name':m.group(1)}; rows.append(cur); continue
    for k in ['VGPRs','AGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','LDS Size \[bytes/block\]','SGPRs','VGPR Spill','SGPR Spill']:
        m=re.search(k+r': (\d+)',l)
        if m and cur is not None: cur[k.split(' ')[0]+('Spill' if 'Spill' in k else '')]=m.group(1)
for r in rows:
    n=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()
    n=re.sub(r'\(.*','',n)[:150]
    print(n, {k:v for k,v in r.items() if k!='name'})
