cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/w16/sq -- $R/tools/wino16_bound > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/w16/grbm -- $R/tools/wino16_bound > /dev/null 2>&1
python - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
def load(d):
    f=glob.glob(os.path.join(R,'gpurun_out/w16',d,'**','*_counter_collection.csv'),recursive=True)[0]
    acc=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
    seen=set()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        acc[k][r['Counter_Name']].append((r['Dispatch_Id'],float(r['Counter_Value'])))
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id']); dur[k].append(float(r['End_Timestamp'])-float(r['Start_Timestamp']))
    return acc,dur
sq,dur=load('sq'); gr,dur2=load('grbm')
for k in sq:
    def avg(c, src): 
        by=collections.defaultdict(float)
        for d,v in src[k][c]: by[d]+=v
        vals=sorted(by.values()); return sum(vals)/len(vals)
    mf=avg('SQ_VALU_MFMA_BUSY_CYCLES',sq); act=avg('GRBM_GUI_ACTIVE',gr)/8.0
    ns=sum(dur2[k])/len(dur2[k])
    print("%-60s %8.3f ms  clock %.2f GHz  MFMA busy %.3f  LDS conflict frac %.3f" % (k.replace('void ',''), ns/1e6, act/ns, mf/(1024.0*act), avg('SQ_LDS_BANK_CONFLICT',sq)/max(avg('SQ_LDS_IDX_ACTIVE',sq),1)))
PY
