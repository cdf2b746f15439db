# development aid: Klein sampler forms (tools/bench_ensdlp.py <tag> <batch> <reps> <mapping>): 0 default, lanes + 0x100 held entries, lanes + 0x200 own-step search
for cfg in "enssig0 4096" "enssig0 16384" "enssig0 65536" "dlpsig0 4096" "dlpsig0 16384" "dlpsig0 65536" "enssig1 16384"; do
  set -- $cfg
  for m in ${KLEIN_MAPS:-0 264 520}; do
    echo "mapping=$m $(python tools/bench_ensdlp.py $1 $2 3 $m 2>&1 | tail -1 | cut -c1-80)"
  done
done
