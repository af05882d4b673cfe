import json,shutil,glob
O='gpurun_out/r01f/'
f=json.load(open(O+'fetch.json')); w=json.load(open(O+'write.json'))
out={"note":"rocprofv3 --pmc <counter> --kernel-trace, one counter per pass (FETCH_SIZE, WRITE_SIZE), `python3 bench.py --cpu-sample 0 --steps 3` (round = 16 x 5 Mbp); values in KB per launch (mean over the launches of the run; max = a full round of 80 M bases against the fullest reference). Calibration (earlier passes, same counters): the 1 GiB hipMemset reads WRITE_SIZE = 1048576 KB exactly; a kernel of 80 M independent 8-byte table gathers read FETCH_SIZE = 5.02e6 KB = 64 B per gather, i.e. one 64-byte request per random read, so no x2 correction applies to these access patterns.","kernels":{}}
for k in sorted(set(f)|set(w)):
    src=f.get(k) or w.get(k)
    e={"launches":src[list(src.keys())[0]]["launches"]}
    if k in f: e["FETCH_SIZE_KB"]={"mean":round(f[k]["FETCH_SIZE"]["mean"],1),"max":round(f[k]["FETCH_SIZE"]["max"],1)}
    if k in w: e["WRITE_SIZE_KB"]={"mean":round(w[k]["WRITE_SIZE"]["mean"],1),"max":round(w[k]["WRITE_SIZE"]["max"],1)}
    out["kernels"][k]=e
json.dump(out,open('profiles/r01_pmc_hbm_traffic.json','w'),indent=1)
shutil.copy(O+'timed.json','profiles/r01_bench_timed_kernel_avgs.json')
ks=glob.glob(O+'kt/**/*kernel_stats.csv',recursive=True)[0]
shutil.copy(ks,'profiles/r01_bench_kernel_stats.csv')
for src,dst in (('bench_full.json','r01_bench_full_path.json'),('bench_matcher.json','r01_bench_matcher_only.json'),('bench_rocprof.json','r01_bench_under_rocprof.json')):
    line=[l for l in open(O+src) if l.startswith('{')][-1]
    open('profiles/'+dst,'w').write(json.dumps(json.loads(line),indent=1)+"\n")
K=out["kernels"]
def mx(n):
    e=K[n]; return e.get("FETCH_SIZE_KB",{}).get("max",0), e.get("WRITE_SIZE_KB",{}).get("max",0)
em=[k for k in K if 'k_emit_' in k]
print("emit kernels",len(em)); print(" + ".join("%.1f + %.1f"%mx(k) for k in em))
for n in K:
    if any(x in n for x in ("resolve_blocks","k_stitch","k_gather","k_copy_multi","k_insert_multi")): print(n,mx(n))
