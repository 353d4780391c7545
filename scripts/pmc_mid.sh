export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $c -d /tmp/p_$c -o b -- python $GRAFT_REPO_ROOT/scripts/time_lde.py 20 > /tmp/p_$c.log 2>&1; done
python - <<'PY'
import sqlite3, glob
for c in ("FETCH_SIZE","WRITE_SIZE"):
    db = sqlite3.connect(sorted(glob.glob(f"/tmp/p_{c}/*.db"))[0])
    tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    pmc=[t for t in tabs if 'pmc_event' in t][0]; disp=[t for t in tabs if 'kernel_dispatch' in t][0]; sym=[t for t in tabs if 'kernel_symbol' in t][0]
    q=f"select s.kernel_name, count(*), avg(p.value) from {pmc} p join {disp} d on p.event_id=d.event_id join {sym} s on d.kernel_id=s.id group by s.kernel_name"
    try:
        for name,n,v in db.execute(q): 
            if 'lde_middle' in name or 'ntt_strided' in name: print(c, name[:70], n, round(v))
    except Exception as e: print('query failed', e, tabs[:20])
PY
