# BAM -> BAM on the GPU box after the zero-copy subread views: the host side alone (--host-only) and end to end
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_cli3; mkdir -p $O
[ -f /tmp/big.subreads.bam ] || timeout 300 $R/ccs_amd/bin/ccs --write-synthetic 32768,10,10000,5 /tmp/big.subreads.bam
for rep in 1 2; do timeout 300 $R/ccs_amd/bin/ccs /tmp/big.subreads.bam --host-only --batch-size 4096 2>&1 | cut -c1-230; done
for w in 3 6; do timeout 300 $R/ccs_amd/bin/ccs /tmp/big.subreads.bam --host-only --batch-size 4096 --workers-per-gpu $w 2>&1 | tail -1 | cut -c1-230; done
for cfg in "4096 3" "4096 3" "4096 6" "8192 4"; do
  set -- $cfg
  timeout 600 $R/ccs_amd/bin/ccs /tmp/big.subreads.bam /tmp/big.hifi.bam --batch-size $1 --workers-per-gpu $2 --log-level INFO > $O/x.log 2>&1
  echo "== batch $1 packers $2: $(grep -E 'ZMWs in' $O/x.log | sed 's/.*out, //') | $(grep -E 'GPU workers' $O/x.log | sed 's/.*: waiting/waiting/') | $(grep -E 'reader thread' $O/x.log | sed 's/.*thread: //')"
done
