"""GPU idle time of one training step (union of all streams) from a rocprofv3 kernel trace, attributed to the kernel that follows each gap."""
import csv,glob,sys,collections
f=(glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv')+glob.glob(sys.argv[1]+'/*_kernel_trace.csv'))[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_im2col_stem' in r['Kernel_Name']]
step=rows[idx[-2]:idx[-1]]
print('step wall', (int(step[-1]['End_Timestamp'])-int(step[0]['Start_Timestamp']))/1e6, 'kernels', len(step))
# union busy of all streams
iv=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in step)
busy=0; cs,ce=iv[0]
gaps=[]
for a,b in iv[1:]:
    if a<=ce: ce=max(ce,b)
    else:
        busy+=ce-cs; gaps.append((a-ce, ce)); cs,ce=a,b
busy+=ce-cs
print('GPU busy (any stream) ms', busy/1e6, 'idle ms', sum(g for g,_ in gaps)/1e6, 'n gaps', len(gaps))
# attribute gaps to the kernel that follows
byname=collections.defaultdict(lambda:[0,0])
ends={}
for g,t in gaps:
    nxt=min((r for r in step if int(r['Start_Timestamp'])>=t+g), key=lambda r:int(r['Start_Timestamp']))
    n=nxt['Kernel_Name'][:70]; byname[n][0]+=g; byname[n][1]+=1
for n,(g,c) in sorted(byname.items(), key=lambda kv:-kv[1][0])[:25]: print(f"{g/1e3:9.1f} us {c:5d}  before {n}")
