#!/bin/bash
# rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE of pass 1 on row-major [N, D] latents, old form (HW == 1
# through the NCHW kernel: lane = token) / row-major form / the NCHW op on the same values.  GPU box; writes gpurun_out/<dir>/
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-flat_pmc}
mkdir -p $O
export DVQ_LIBRARY=$R/dynamicvectorquantization_amd/csrc/libdvq_tuning.so
cd /tmp && export TMPDIR=/tmp
for f in ref nchw flat; do
  FLAT_PROBE_ONE=$f timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$f -o t -- python3 $R/tools/flat_probe.py > $O/trace_$f.json 2>> $O/err.log
  FLAT_PROBE_ONE=$f timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$f -o p -- python3 $R/tools/flat_probe.py > /dev/null 2>> $O/err.log
  FLAT_PROBE_ONE=$f timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$f -o p -- python3 $R/tools/flat_probe.py > /dev/null 2>> $O/err.log
done
cd $R
python3 - <<PY
import csv, glob, json, collections
O="$O"
cal=json.load(open("$R/profiles/pmc_traffic.json"))["calibration"]["fetch_correction_factor"]
out={"fetch_correction_factor_from_profiles_pmc_traffic_json":cal,"forms":{}}
def counter(d, name):
    fs=glob.glob(d+"/*counter_collection.csv")+glob.glob(d+"/*/*counter_collection.csv")
    per=collections.defaultdict(list)
    if not fs: return per
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"]==name: per[r["Kernel_Name"].split("(")[0].replace("void ","")].append(float(r["Counter_Value"]))
    return per
for f in ("ref","nchw","flat"):
    ent={}
    fs=glob.glob(O+"/trace_%s/*kernel_stats.csv"%f)+glob.glob(O+"/trace_%s/*/*kernel_stats.csv"%f)
    if fs:
        for r in csv.DictReader(open(fs[0])):
            n=r["Name"].replace("void ","").split("(")[0]
            if n.startswith("vq_assign_filter") or n.startswith("vq_resolve") or n.startswith("vq_assign_exact"):
                ent.setdefault("kernel_avg_us",{})[n]=round(float(r["AverageNs"])/1e3,1)
    fe,wr=counter(O+"/fetch_"+f,"FETCH_SIZE"),counter(O+"/write_"+f,"WRITE_SIZE")
    for k in fe:
        if k.startswith("vq_assign_filter"):
            v=fe[k]; w=wr.get(k,[0])
            med=lambda a: sorted(a)[len(a)//2]
            ent["pass1_fetch_MB_corrected"]=round(med(v)*1024*cal/1e6,1); ent["pass1_write_MB"]=round(med(w)*1024/1e6,1); ent["pass1_kernel"]=k
    out["forms"][f]=ent
json.dump(out,open(O+"/flat_pmc.json","w"),indent=1); print(json.dumps(out,indent=1))
PY
