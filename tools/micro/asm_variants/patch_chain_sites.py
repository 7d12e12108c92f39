import re,sys
src,dst,variant=sys.argv[1:4]
lines=open(src).readlines()
# locate chain sites in cv1_rr<16,*,HALF>
sites=[]  # (line index, N, kernel)
name=None;nbar=0;stored=False
for i,l in enumerate(lines):
    m=re.match(r'^(_Z\w+):',l)
    if m: name=m.group(1) if re.search(r'cv1_rr_kernelILi16ELb[01]ELi1E',m.group(1)) else None; nbar=0; stored=False; k=0
    s=l.strip()
    if not name: continue
    if s=='s_barrier': nbar+=1
    if s.startswith(('global_store','buffer_store')): stored=True
    if s.startswith('s_waitcnt') and nbar>0 and not stored:
        m=re.search(r'lgkmcnt\((\d+)\)',s)
        if m and int(m.group(1))>0:
            sites.append((i,int(m.group(1)),name,k)); k+=1
def nextmfma(i):
    for j in range(i+1,i+6):
        t=lines[j].strip()
        if t.startswith('v_mfma'): return t.split()[0]
    return ''
count=0
for (i,N,nm,k) in sites:
    pick=False
    if variant.startswith('S'):      # S<lo>_<hi>: site index range
        lo,hi=map(int,variant[1:].split('_')); pick = lo<=k<hi
    elif variant=='N1': pick=N==1
    elif variant=='N2': pick=N==2
    elif variant=='N34': pick=N>=3
    elif variant=='T': pick='16x16x16' in nextmfma(i)
    elif variant=='X': pick='16x16x32' in nextmfma(i)
    elif variant=='Q': lines[i]=re.sub(r'lgkmcnt\(\d+\)','lgkmcnt(%d)'%(N-1),lines[i]); count+=1; continue
    elif variant=='P': lines[i]=lines[i]+'\ts_nop 7\n'; count+=1; continue
    if pick: lines[i]=re.sub(r'lgkmcnt\(\d+\)','lgkmcnt(0)',lines[i]); count+=1
open(dst,'w').writelines(lines)
print(variant,'patched',count,'of',len(sites))
if variant=='LIST':
    for (i,N,nm,k) in sites:
        if 'Lb0' in nm: print(k,i+1,N,nextmfma(i))
