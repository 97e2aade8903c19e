# the ccs BAM -> BAM path on one MI355X: throughput and where the host time goes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_cli; rm -rf $O; mkdir -p $O
timeout 300 $R/ccs_amd/bin/ccs --write-synthetic 32768,10,10000,5 /tmp/big.subreads.bam
ls -la /tmp/big.subreads.bam* > $O/cli.txt
for bs in 4096 8192; do
( time timeout 600 $R/ccs_amd/bin/ccs /tmp/big.subreads.bam /tmp/big.hifi.bam --batch-size $bs --log-level INFO ) > $O/cli_$bs.log 2>&1
echo "== batch-size $bs" >> $O/cli.txt; grep -E "reader thread|GPU workers|ZMWs in|real" $O/cli_$bs.log >> $O/cli.txt
done
( time timeout 600 $R/ccs_amd/bin/ccs /tmp/big.subreads.bam /tmp/big2.hifi.bam --chunk 2/4 --batch-size 4096 --log-level INFO ) > $O/cli_chunk.log 2>&1
echo "== chunk 2/4" >> $O/cli.txt; grep -E "chunk|ZMWs in|real" $O/cli_chunk.log >> $O/cli.txt
cat $O/cli.txt
