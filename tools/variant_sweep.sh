# usage (on the GPU box): bash tools/variant_sweep.sh <layer name> ...; every tile variant of the convolution kernel on the named encoder layers
for l in "$@"; do
  echo "== $l (heuristic first)"
  timeout -k 10 100 python tools/igemm_sweep.py $l 2>&1 | grep "^$l"
  for f in 0 1 3 8 9 10 11 12 13 14 15 16; do
    printf "v%-3s" $f; MIC_IGEMM_FORCE=$f timeout -k 10 100 python tools/igemm_sweep.py $l 2>&1 | grep "^$l" || echo
  done
done
