import csv, glob, collections, sys
tag=sys.argv[1]; kern=sys.argv[2]
vals={}
for d in ('1','2'):
    for f in glob.glob('/root/repo/gpurun_out/pmc_%s_%s/*/*counter_collection.csv'%(tag,d)):
        acc=collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if kern in row['Kernel_Name']:
                acc[row['Counter_Name']].append(float(row['Counter_Value']))
        for k,v in acc.items(): vals[k]=sum(v)/len(v)
wc=vals['SQ_WAVE_CYCLES']
print("waves %.0f  wave_quadcycles %.3g  per-wave cycles %.0f"%(vals['SQ_WAVES'], wc, 4*wc/vals['SQ_WAVES']))
for k in ('SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_ANY','SQ_WAIT_INST_LDS'): print("  %-20s %.1f%%"%(k,100*vals[k]/wc))
cyc=vals['GRBM_GUI_ACTIVE']/8
print("gpu cycles %.3g"%cyc)
print("VALU insts %.3g -> SIMD busy %.1f%%"%(vals['SQ_INSTS_VALU'], 100*vals["SQ_ACTIVE_INST_VALU"]*4/1024/cyc))
print("LDS insts %.3g, LDS idx active/CU %.1f%% , bank conflict share %.1f%%"%(vals['SQ_INSTS_LDS'],100*vals['SQ_LDS_IDX_ACTIVE']/256/cyc, 100*vals['SQ_LDS_BANK_CONFLICT']/vals['SQ_LDS_IDX_ACTIVE']))
print("VMEM rd %.3g wr %.3g SALU %.3g"%(vals['SQ_INSTS_VMEM_RD'],vals['SQ_INSTS_VMEM_WR'],vals['SQ_INSTS_SALU']))
