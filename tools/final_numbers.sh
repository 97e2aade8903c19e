# end-of-round numbers besides the default bench: the other BASELINE shapes and the ccs BAM -> BAM path.  usage: bash tools/final_numbers.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-final}; O=$R/gpurun_out/$T; rm -rf $O; mkdir -p $O
for w in c1 c4 c5; do timeout 500 python $R/bench.py --workload $w --steps 4 --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err; done
timeout 300 $R/ccs_amd/bin/ccs --write-synthetic 32768,10,10000,5 /tmp/big.subreads.bam
ls -la /tmp/big.subreads.bam* > $O/cli.txt
( time timeout 600 $R/ccs_amd/bin/ccs /tmp/big.subreads.bam /tmp/big.hifi.bam --batch-size 4096 --log-level INFO ) > $O/cli.log 2>&1
tail -8 $O/cli.log >> $O/cli.txt; ls -la /tmp/big.hifi.bam* >> $O/cli.txt
( time timeout 600 $R/ccs_amd/bin/ccs /tmp/big.subreads.bam /tmp/big2.hifi.bam --chunk 2/4 --batch-size 4096 ) >> $O/cli.txt 2>&1
python - <<PY
import json
for w in ("c1","c4","c5"):
    d=json.loads(open("$O/bench_%s.json"%w).read().strip().splitlines()[-1])
    print(w, d["value"], d["resident_zmws_per_s"], d["config"]["zmws_per_gpu"], d["stage_ms"]["total_ms"], d["cpu_baseline"]["value"], d["cpu_baseline"].get("gpu_matches_cpu_sequences"), d.get("success_frac"))
PY
cat $O/cli.txt
