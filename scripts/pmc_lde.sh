#!/bin/bash
# PMC counters of the commit-stage kernels (scripts/time_lde.py as the workload).  Separate passes, --kernel-trace only.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_lde; rm -rf $OUT; mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS -d $OUT/a -o p -- python $R/scripts/time_lde.py ${1:-20} > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVES -d $OUT/b -o p -- python $R/scripts/time_lde.py ${1:-20} > $OUT/b.log 2>&1
python - <<PY
import sqlite3, glob
for d in ("a", "b"):
    f = sorted(glob.glob("$OUT/" + d + "/*.db"))
    if not f: print("no db for", d); continue
    c = sqlite3.connect(f[0])
    q = "select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection group by kernel_name, counter_name"
    rows = {}
    for name, cn, v, n, dur in c.execute(q):
        s = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
        if not any(k in s for k in ("lde_", "ntt_", "leaf_hash", "main_trace")): continue
        rows.setdefault(s, {"dur_us": dur / 1e3})[cn] = v
    for s, r in rows.items():
        print(s, {k: (round(v, 1) if k == "dur_us" else float(f"{v:.4g}")) for k, v in r.items()})
PY
