# SPEC approximation studies on on-model, off-model and low-complexity data (oracle, CPU): every build-defined approximation against its
# exact / wide counterpart.  usage: bash tools/spec_studies.sh > profiles/r04_spec_studies.txt
cd "$(dirname "$0")/.."
echo "# consensus errors against the true templates (edit distance), mean rq, polish rounds per window; tools/acc_eval.py N PASSES LENGTH SEED key=value"
echo "# data sets: on-model = the library's generator (SURVEY.md 8d channel); channel=1.5 / 0.5 = every error rate scaled; hp_boost=2.5 = indels 2.5x inside"
echo "# homopolymers (a context dependence the SYN-1 parameter set does not have); tpl=lowcx = homopolymer runs 5-60 + 2-4-mer tandem repeats 20-500 bp"
for data in "" "channel=1.5" "hp_boost=2.5" "tpl=lowcx"; do
  for P in 10 5; do
    echo "## data: ${data:-on-model}  passes $P x 5 kb, 48 ZMWs"
    for knob in "" "poa_band=64" "align_band1=64" "sat_rows=0 sat_gain=-99999999" "sat_rows=2" "sat_rows=0" "score_band=64" "score_band=4" "skip_margin=4" "skip_margin=8" "disable_heuristics=1" "min_zscore=0"; do
      python tools/acc_eval.py 48 $P 5000 $((40 + P)) $data $knob | cut -c1-230
    done
  done
done
