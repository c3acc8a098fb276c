import itertools, random
M=0xFFFF
def lo(x): return x & M
def hi(x): return x >> 16
def pk(a,b): return (a&M)|((b&M)<<16)
def pkmin(x,y): return pk(min(lo(x),lo(y)), min(hi(x),hi(y)))
def pkmax(x,y): return pk(max(lo(x),lo(y)), max(hi(x),hi(y)))
def rot(x): return pk(hi(x), lo(x))
def lolo(a,b): return pk(lo(a), lo(b))   # perm sel 0x05040100 with s0=b s1=a
def hihi(a,b): return pk(hi(a), hi(b))
NET8=[(0,1),(2,3),(4,5),(6,7),(0,2),(1,3),(4,6),(5,7),(1,2),(5,6),(0,4),(1,5),(2,6),(3,7),(2,4),(3,5),(1,2),(3,4),(5,6)]
def sort16(d):
    d=list(d); ops=0
    for i,j in NET8:
        t=pkmin(d[i],d[j]); d[j]=pkmax(d[i],d[j]); d[i]=t; ops+=2
    a=[0]*4; b=[0]*4
    for i in range(4):
        s=rot(d[7-i]); a[i]=pkmin(d[i],s); b[i]=pkmax(d[i],s); ops+=3
    # a[i]=(X_i,X_{7-i}), b[i]=(Y_i,Y_{7-i}); each 8-seq bitonic
    out=[]
    for q in (a,b):
        # distance 4
        lowq=[0,0]; highq=[0,0]
        for i in range(2):
            s=rot(q[3-i]); lowq[i]=pkmin(q[i],s); highq[i]=pkmax(q[i],s); ops+=3
        # lowq[0]=(p0,p3) lowq[1]=(p1,p2); highq[0]=(p4,p7) highq[1]=(p5,p6)
        for r0,r1 in (lowq,highq):
            s=rot(r1); mn=pkmin(r0,s); mx=pkmax(r0,s); ops+=3
            # mn=(pos0,pos1) mx=(pos2,pos3) of this quad, each pair bitonic(any order): distance-1 stage
            t1=lolo(mn,mx); t2=hihi(mn,mx); ops+=2      # t1=(pos0,pos2) t2=(pos1,pos3)
            m2=pkmin(t1,t2); x2=pkmax(t1,t2); ops+=2    # m2=(min01,min23) x2=(max01,max23)
            out.append(lolo(m2,x2)); out.append(hihi(m2,x2)); ops+=2
    return out, ops
def flat(d): 
    r=[]
    for x in d: r+= [lo(x),hi(x)]
    return r
# 0-1 principle: all 2^16 binary inputs
bad=0
for bits in range(1<<16):
    e=[(bits>>i)&1 for i in range(16)]
    d=[pk(e[2*i],e[2*i+1]) for i in range(8)]
    o,ops=sort16(d)
    f=flat(o)
    if f!=sorted(e): bad+=1
print("bad",bad,"ops",ops)
random.seed(1)
for _ in range(2000):
    e=[random.randrange(65536) for _ in range(16)]
    d=[pk(e[2*i],e[2*i+1]) for i in range(8)]
    o,_=sort16(d)
    assert flat(o)==sorted(e)
print("random ok")

def merge16(d):
    d=list(d)
    a=[0]*4; b=[0]*4
    for i in range(4):
        s=rot(d[7-i]); a[i]=pkmin(d[i],s); b[i]=pkmax(d[i],s)
    out=[]
    for q in (a,b):
        lowq=[0,0]; highq=[0,0]
        for i in range(2):
            s=rot(q[3-i]); lowq[i]=pkmin(q[i],s); highq[i]=pkmax(q[i],s)
        for r0,r1 in (lowq,highq):
            s=rot(r1); mn=pkmin(r0,s); mx=pkmax(r0,s)
            t1=lolo(mn,mx); t2=hihi(mn,mx)
            m2=pkmin(t1,t2); x2=pkmax(t1,t2)
            out.append(lolo(m2,x2)); out.append(hihi(m2,x2))
    return out
for _ in range(20000):
    A=sorted(random.randrange(65536) for _ in range(8)); B=sorted(random.randrange(40000,65536) for _ in range(8))
    a=[pk(A[2*i],A[2*i+1]) for i in range(4)]; b=[pk(B[2*i],B[2*i+1]) for i in range(4)]
    d=[0]*8
    for m in range(4):
        d[2*m]=lolo(a[m],b[m]); d[2*m+1]=hihi(a[m],b[m])
    assert flat(merge16(d))==sorted(A+B)
print("merge of two runs ok")
