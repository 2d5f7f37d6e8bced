"""Per-block spans of one training step from a rocprofv3 kernel trace (main stream only):
    rocprofv3 --kernel-trace --output-format csv -d OUT -o r -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline
    python tools/phase_spans.py OUT"""
import csv,glob,collections,sys
f=(glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv')+glob.glob(sys.argv[1]+'/*_kernel_trace.csv'))[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_im2col_stem' in r['Kernel_Name']]
step=rows[idx[-1]:]
main_stream=step[0]['Stream_Id']
fwd_bounds=[12,13,37,38,110,111,159]; fwd_names=['e2','t2','e3','t3','e4','t4','dec']
bwd_bounds=[48,49,121,122,146,147,159]; bwd_names=['dec_b','t4_b','e4_b','t3_b','e3_b','t2_b','e2_b']
spans=collections.OrderedDict(); nf=0; nb=0; inbwd=False
for r in step:
    if r['Stream_Id']!=main_stream: continue
    n=r['Kernel_Name']; st=int(r['Start_Timestamp']); en=int(r['End_Timestamp'])
    if 'k_nchw_to_nhwc' in n: inbwd=True
    if not inbwd: phase='head+loss' if nf>=159 else next(nm for bnd,nm in zip(fwd_bounds,fwd_names) if nf<bnd)
    else: phase='stem_b+opt' if nb>=159 else next(nm for bnd,nm in zip(bwd_bounds,bwd_names) if nb<bnd)
    a=spans.setdefault(phase,[st,en,0.0]); a[0]=min(a[0],st); a[1]=max(a[1],en); a[2]+=(en-st)/1e6
    if 'k_bn_finalize' in n: nf+=1
    if 'k_bn_bwd_coeffs' in n or 'k_bn_bwd_apply' in n: nb+=1
for k,(a,b,t) in spans.items(): print(f"{k:12s} span {(b-a)/1e6:7.2f} ms  main-stream kernel-sum {t:7.2f} ms")
print('wall', (int(step[-1]['End_Timestamp'])-int(step[0]['Start_Timestamp']))/1e6)
