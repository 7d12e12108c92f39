import re,sys
src,dst,variant=sys.argv[1:4]
out=[];name=None;nbar=0;stored=False;count=0
for l in open(src):
    m=re.match(r'^(_Z\w+):',l)
    if m: name=m.group(1) if re.search(r'cv1_rr_kernelILi16ELb[01]ELi1E',m.group(1)) else None; nbar=0; stored=False
    s=l.strip()
    if name:
        if s=='s_barrier': nbar+=1
        if s.startswith('global_store') or s.startswith('buffer_store'): stored=True
        if s.startswith('s_endpgm'): name_end=True
        if s.startswith('s_waitcnt') and 'ASM' not in s:
            region='pro' if nbar==0 else ('tail' if stored else 'chain')
            new=None
            if variant=='A': new='s_waitcnt vmcnt(0) lgkmcnt(0)'
            elif variant=='B' and 'lgkmcnt' in s: new=re.sub(r'lgkmcnt\(\d+\)','lgkmcnt(0)',s)
            elif variant=='C' and 'vmcnt' in s: new=re.sub(r'vmcnt\(\d+\)','vmcnt(0)',s)
            elif variant=='D' and region=='pro': new='s_waitcnt vmcnt(0) lgkmcnt(0)'
            elif variant=='E' and region=='chain': new='s_waitcnt vmcnt(0) lgkmcnt(0)'
            elif variant=='F' and region=='chain' and 'lgkmcnt' in s: new=re.sub(r'lgkmcnt\(\d+\)','lgkmcnt(0)',s)
            elif variant=='G' and region=='chain' and 'vmcnt' in s: new=re.sub(r'vmcnt\(\d+\)','vmcnt(0)',s)
            elif variant=='H' and region=='tail': new='s_waitcnt vmcnt(0) lgkmcnt(0)'
            if new and new!=s: l='\t'+new+'\n'; count+=1
    out.append(l)
open(dst,'w').writelines(out)
print(variant,'patched',count)
