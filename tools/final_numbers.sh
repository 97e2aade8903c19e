# end-of-round numbers besides the default bench (whose `extra` carries the other BASELINE shapes): the `ccs` driver BAM -> BAM, incl. a configs[4]-like mix through
# cost-binned batches.  usage (through gpurun): bash tools/final_numbers.sh <tag>   -> gpurun_out/<tag>/cli.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-final}; O=$R/gpurun_out/$T; mkdir -p $O
CCS=$R/ccs_amd/bin/ccs
{
echo "== 32768 ZMWs x 10 passes x 10 kb (4.5 GB BAM)"
timeout 300 $CCS --write-synthetic 32768,10,10000,5 /tmp/big.subreads.bam; ls -la /tmp/big.subreads.bam | awk '{print $5, $9}'
timeout 300 $CCS --host-only /tmp/big.subreads.bam 2>&1 | tail -1
( time timeout 600 $CCS /tmp/big.subreads.bam /tmp/big.hifi.bam --batch-size 4096 --log-level INFO ) 2>&1 | tail -6
echo "== 8192 ZMWs, 3-50 passes x 1-25 kb (configs[4] shape), --min-rq 0.99, cost-binned batches"
timeout 300 $CCS --write-synthetic 8192,3-50,1000-25000,9 /tmp/mix.subreads.bam; ls -la /tmp/mix.subreads.bam | awk '{print $5, $9}'
timeout 300 $CCS --host-only --batch-size 4096 /tmp/mix.subreads.bam 2>&1 | tail -1
( time timeout 600 $CCS /tmp/mix.subreads.bam /tmp/mix.hifi.bam --min-rq 0.99 --log-level INFO ) 2>&1 | tail -6        # (default --batch-size 2048, --batch-bases 110 kb x 2048)
( time timeout 600 $CCS /tmp/mix.subreads.bam /tmp/mix2.hifi.bam --batch-size 4096 --batch-bases 800000000 --min-rq 0.99 ) 2>&1 | tail -4   # (round 4's cut: two tickets)
cmp /tmp/mix.hifi.bam /tmp/mix2.hifi.bam && echo "hifi.bam identical for both batch cuts"
} > $O/cli.txt 2>&1
cat $O/cli.txt
