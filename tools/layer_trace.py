"""Timeline (stream, start, duration, kernel) of two dense_e4 layers in forward and in backward from a rocprofv3 kernel trace of bench.py."""
import csv,glob,sys
f=(glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv')+glob.glob(sys.argv[1]+'/*_kernel_trace.csv'))[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_im2col_stem' in r['Kernel_Name']]
step=rows[idx[-2]:idx[-1]]
main=step[0]['Stream_Id']
def short(n):
    n=n.replace('void ','').replace('rdm::','')
    return n[:62]
# forward: find finalize count to locate e4 layer 20 (finalize index: e2 12, t2 1, e3 24, t3 1 -> e4 starts at 38; layer 20 -> 38+40)
nf=0; start=None; end=None
for i,r in enumerate(step):
    if r['Stream_Id']==main and 'k_bn_finalize' in r['Kernel_Name']:
        nf+=1
        if nf==38+40+1 and start is None: start=i
        if nf==38+44+1: end=i; break
t0=int(step[start]['Start_Timestamp'])
print("---- forward, dense_e4 layers 21-22 ----")
for r in step[start:end]:
    print(f"{'M' if r['Stream_Id']==main else 'S'} +{(int(r['Start_Timestamp'])-t0)/1e3:8.1f} us  dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f}  {short(r['Kernel_Name'])}")
# backward: locate by k_bn_bwd_apply count: dec 48, t4: coeffs 1, then e4: layer from the end
nb=0; start=None; end=None
for i,r in enumerate(step):
    if r['Stream_Id']==main and ('k_bn_bwd_apply' in r['Kernel_Name'] or 'k_bn_bwd_coeffs' in r['Kernel_Name']):
        nb+=1
        if nb==49+30 and start is None: start=i
        if nb==49+34: end=i; break
t0=int(step[start]['Start_Timestamp'])
print("---- backward, two dense_e4 layers ----")
for r in step[start:end]:
    print(f"{'M' if r['Stream_Id']==main else 'S'} +{(int(r['Start_Timestamp'])-t0)/1e3:8.1f} us  dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f}  {short(r['Kernel_Name'])}")
