for cfg in "dlpsig0 16384 4" "dlpsig0 16384 8" "dlpsig0 16384 16" "dlpsig0 8192 4" "dlpsig0 8192 8" "dlpsig0 8192 16" "enssig0 8192 8" "enssig0 8192 16" "enssig0 16384 16" "enssig0 32768 8" "enssig0 32768 16" "enssig1 65536 16" "enssig1 4096 16"; do
  set -- $cfg
  echo "lanes=$3 $(python tools/bench_ensdlp.py $1 $2 3 $3 2>&1 | tail -1 | cut -c1-80)"
done
