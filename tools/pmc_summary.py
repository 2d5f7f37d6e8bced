"""One line per conv kernel from a rocprofv3 --pmc SQ_* counter collection (see profiles/r01_pmc_sq_conv_kernels_e2.txt for the command)."""
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
seen={}
for r in rows:
    if any(k in r['Kernel_Name'] for k in ('conv_','conv3x3','gemm_bf16','gemm_panel')):
        d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
        seen.setdefault((r['Dispatch_Id'],r['Kernel_Name'][10:58],r['Grid_Size'],r['VGPR_Count'],r['LDS_Block_Size']),{'dur':d})[r['Counter_Name']]=float(r['Counter_Value'])
last={}
for k,v in seen.items(): last[k[1]]=(k,v)
for name,(k,v) in last.items():
    g=lambda c: v.get(c,0.0)
    wc=g('SQ_WAVE_CYCLES') or 1
    extra = ' '.join(f"{c}={g(c)/1e6:.1f}M" for c in sorted(v) if c not in ('dur','SQ_WAVES','SQ_WAVE_CYCLES','SQ_BUSY_CYCLES','SQ_VALU_MFMA_BUSY_CYCLES','SQ_ACTIVE_INST_ANY','SQ_LDS_IDX_ACTIVE','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_LDS_BANK_CONFLICT'))
    print(name, extra, f"grid {k[2]} vgpr {k[3]} lds {k[4]} dur {v['dur']:.0f}us clk {g('SQ_BUSY_CYCLES')/32/v['dur']/1e3:.2f}GHz waves {g('SQ_WAVES'):.0f} life/wave {4*wc/max(g('SQ_WAVES'),1)/1e6:.3f}Mcyc wait_any {g('SQ_WAIT_ANY')/wc:.2f} wait_inst {g('SQ_WAIT_INST_ANY')/wc:.2f} active {g('SQ_ACTIVE_INST_ANY')/wc:.2f} mfma_busy/busy {g('SQ_VALU_MFMA_BUSY_CYCLES')/1024/(g('SQ_BUSY_CYCLES')/32+1):.2f} lds_act {g('SQ_LDS_IDX_ACTIVE')/1e6:.0f}M conflict {g('SQ_LDS_BANK_CONFLICT')/1e6:.0f}M")
