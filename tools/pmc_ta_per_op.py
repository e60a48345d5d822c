"""Per-op texture-addresser (TA) load of the network: TA_TA_BUSY, TA address / data stalls by the cache, L1 -> L2 read requests (separate --pmc passes of
`REQ_COUNTERS="TA_TA_BUSY_sum ... GRBM_GUI_ACTIVE" tools/gpu_run.sh <dir> pmcreq`, single lane) next to each op's duration.

    python tools/pmc_ta_per_op.py PMC_DIR [VARIANT BATCH]
"""
import csv,sys,collections
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from head_detector_amd import arch
d=sys.argv[1]
variant, batch = (sys.argv[2], int(sys.argv[3])) if len(sys.argv) > 3 else ('vgg_heads_l', 64)
def load(name):
    rows=[r for r in csv.DictReader(open(f'{d}/{name}_counter_collection.csv')) if arch.is_net_kernel(r["Kernel_Name"]) and r["Counter_Name"]==name]
    rows.sort(key=lambda r:int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rows], [r["Kernel_Name"] for r in rows], [(int(r["End_Timestamp"])-int(r["Start_Timestamp"])) for r in rows]
P=arch.build_program(variant,arch.random_state_dict(variant,1),640)
n=len([o for o in P.ops if o['kind'] in (0,1,2)])
ops=[o for o in P.ops if o['kind'] in (0,1,2)]
C={}
for c in ("TA_TA_BUSY_sum","TA_BUFFER_TOTAL_CYCLES_sum","TA_ADDR_STALLED_BY_TC_CYCLES_sum","TA_DATA_STALLED_BY_TC_CYCLES_sum","TCP_TCC_READ_REQ_sum","TCP_PENDING_STALL_CYCLES_sum","GRBM_GUI_ACTIVE","TA_BUFFER_READ_LDS_WAVEFRONTS_sum"):
    v,k,t=load(c); C[c]=v[n:2*n]; 
    if c=="GRBM_GUI_ACTIVE": kn=k[n:2*n]; dur=t[n:2*n]
print(len(C["GRBM_GUI_ACTIVE"]),n)
print(f"{'op':40s} {'us':>6s} {'TAbusy%':>7s} {'addrStallTC%':>8s} {'dataStallTC%':>8s} {'rdreq/us':>9s} {'B/req':>6s}  kernel")
for i,op in enumerate(ops):
    act=C["GRBM_GUI_ACTIVE"][i]/8  # cycles per XCD
    ta=C["TA_TA_BUSY_sum"][i]/(act*256) if act else 0
    a_st=C["TA_ADDR_STALLED_BY_TC_CYCLES_sum"][i]/(act*256) if act else 0
    d_st=C["TA_DATA_STALLED_BY_TC_CYCLES_sum"][i]/(act*256) if act else 0
    by=arch.op_algorithmic_bytes(P,op,batch)
    req=C["TCP_TCC_READ_REQ_sum"][i]
    us=dur[i]/1e3
    k=kn[i].replace("void (anonymous namespace)::","")[:44]
    if us>40: print(f"{op['name'][:40]:40s} {us:6.0f} {100*ta:7.1f} {100*a_st:8.1f} {100*d_st:8.1f} {req/us:9.0f} {by['read']/max(req,1):6.0f}  {k}")
