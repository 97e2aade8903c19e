# BAM -> BAM at a size where the start-up (page-locking the staging arenas) stops dominating
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_cli3; mkdir -p $O
df -h /tmp | tail -1
timeout 600 $R/ccs_amd/bin/ccs --write-synthetic 98304,10,10000,5 /tmp/huge.subreads.bam; ls -la /tmp/huge.subreads.bam
timeout 300 $R/ccs_amd/bin/ccs /tmp/huge.subreads.bam --host-only --batch-size 4096 2>&1 | tail -1 | cut -c1-230
for cfg in "4096 3" "4096 3" "2048 3"; do
  set -- $cfg
  timeout 600 $R/ccs_amd/bin/ccs /tmp/huge.subreads.bam /tmp/huge.hifi.bam --batch-size $1 --workers-per-gpu $2 --log-level INFO > $O/x.log 2>&1
  echo "== batch $1 packers $2: $(grep -E 'ZMWs in' $O/x.log | sed 's/.*out, //') | $(grep -E 'GPU workers' $O/x.log | sed 's/.*: waiting/waiting/') | $(grep -E 'reader thread' $O/x.log | sed 's/.*thread: //')"
done
grep -i "eta\|progress" $O/x.log | tail -3
