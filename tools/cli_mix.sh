# the `ccs` driver BAM -> BAM on a LARGER configs[4]-like mix (the 8192-ZMW file of tools/final_numbers.sh is over in 2 s: start-up dominates it)
# usage (through gpurun): bash tools/cli_mix.sh <tag> [n_zmws]  -> gpurun_out/<tag>/cli_mix.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-final}; N=${2:-24576}; O=$R/gpurun_out/$T; mkdir -p $O
CCS=$R/ccs_amd/bin/ccs
{
echo "== $N ZMWs, 3-50 passes x 1-25 kb (configs[4] shape), --min-rq 0.99, default batch cut (--batch-size 2048, --batch-bases 110 kb x 2048)"
timeout 600 $CCS --write-synthetic $N,3-50,1000-25000,9 /tmp/mixL.subreads.bam; ls -la /tmp/mixL.subreads.bam | awk '{print $5, $9}'
timeout 300 $CCS --host-only /tmp/mixL.subreads.bam 2>&1 | tail -1
( time timeout 900 $CCS /tmp/mixL.subreads.bam /tmp/mixL.hifi.bam --min-rq 0.99 --log-level INFO ) 2>&1 | tail -6
( time timeout 900 $CCS /tmp/mixL.subreads.bam /tmp/mixL2.hifi.bam --min-rq 0.99 --batch-size 4096 ) 2>&1 | tail -4
cmp /tmp/mixL.hifi.bam /tmp/mixL2.hifi.bam && echo "hifi.bam identical for both batch cuts"
rm -f /tmp/mixL*.bam
} > $O/cli_mix.txt 2>&1
cat $O/cli_mix.txt
