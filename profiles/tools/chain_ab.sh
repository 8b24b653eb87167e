# A/B of the chain kernel (launches 1..nb as one persistent kernel) against a kernel per launch
for spec in "c1 1" "c2 1" "c2 8" "c4 1" "c4 8"; do
  set -- $spec
  for t in "" "no_chain=1" "chain_from=2"; do
    HRT_TUNE=$t python profiles/tools/step_ms.py $1 $2 2>/dev/null
  done
done
