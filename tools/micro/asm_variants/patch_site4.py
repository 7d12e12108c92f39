import re,sys
src,dst,variant=sys.argv[1:4]
lines=open(src).readlines()
name=None;nbar=0;stored=False;k=0;count=0
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
            if k==4:
                count+=1
                if variant=='S4': lines[i]='\ts_waitcnt lgkmcnt(0)\n'
                elif variant=='G0': lines[i]=l+'\ts_nop 0\n'
                elif variant=='G1': lines[i]=l+'\ts_nop 1\n'
                elif variant=='G3': lines[i]=l+'\ts_nop 3\n'
                elif variant=='G7': lines[i]=l+'\ts_nop 7\n'
                elif variant=='SP':   # split the read2 into two ds_read_b64 and count accordingly
                    j=i-1
                    while 'ds_read2st64_b64' not in lines[j]: j-=1
                    mm=re.match(r'\s*ds_read2st64_b64 v\[(\d+):(\d+)\], (v\d+) offset0:(\d+) offset1:(\d+)',lines[j])
                    a=int(mm.group(1)); base=mm.group(3); o0=int(mm.group(4))*512; o1=int(mm.group(5))*512
                    lines[j]='\tds_read_b64 v[%d:%d], %s offset:%d\n\tds_read_b64 v[%d:%d], %s offset:%d\n'%(a,a+1,base,o0,a+2,a+3,base,o1)
                    w=[t for t in range(j+1,i) if 'lgkmcnt' in lines[t]]
                    assert len(w)==1 and 'lgkmcnt(1)' in lines[w[0]], [lines[t] for t in w]
                    lines[w[0]]='\ts_waitcnt lgkmcnt(2)\n'
                    lines[i]='\ts_waitcnt lgkmcnt(2)\n'
                elif variant=='NB':   # a v_nop-like independent VALU instead of time: does ANY instruction in between help?
                    lines[i]=l+'\tv_mov_b32 v70, v70\n'
            k+=1
open(dst,'w').writelines(lines); print(variant,'patched',count)
