cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_cli; mkdir -p $O
[ -f /tmp/big.subreads.bam ] || timeout 300 $R/ccs_amd/bin/ccs --write-synthetic 32768,10,10000,5 /tmp/big.subreads.bam
for cfg in "two 4096" "serial 4096" "two 2048" "serial 8192" "two 4096"; do
  set -- $cfg
  env CCSX_SERIAL_STAGES=$([ $1 = serial ] && echo 1 || echo 0) timeout 600 $R/ccs_amd/bin/ccs /tmp/big.subreads.bam /tmp/big.hifi.bam --batch-size $2 --log-level INFO > $O/x.log 2>&1
  echo "== $cfg: $(grep -E 'ZMWs in' $O/x.log | sed 's/.*out, //') | $(grep -E 'GPU workers' $O/x.log | sed 's/.*: waiting/waiting/') | $(grep -E 'reader thread' $O/x.log | sed 's/.*thread: //')"
done
