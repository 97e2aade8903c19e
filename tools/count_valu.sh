# attribute k_polish's VALU instructions to its parts: counting runs with parts of the kernel compiled out (results are garbage)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "" _nofill _noscore; do
  rm -rf /tmp/cv; CCSX_LIB=$R/ccs_amd/libccsx$v.so timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d /tmp/cv -o pmc -- python $R/tools/phase_run.py > /dev/null 2>&1
  python - <<PY
import sqlite3,glob
db=glob.glob('/tmp/cv/**/pmc_results.db',recursive=True)[0]
c=sqlite3.connect(db)
for kn,cn,v,n in c.execute("select kernel_name,counter_name,sum(value),count(*) from counters_collection where kernel_name like 'k_polish%' group by kernel_name,counter_name"):
    print("lib$v", cn, int(v/n/4096), "per ZMW")
PY
done
