# the fuzzers at ten times fuzz_all.sh's case counts under other seeds (a round's last long check): fuzz_long.sh [seed offset]
o=${1:-100}
for f in "fuzz_generic.py 500 $((o+1))" "fuzz_pipeline.py 100 $((o+2))" "fuzz_stream.py 200 $((o+3))" "fuzz_train.py 150 $((o+4))" "fuzz_mfcc.py 150 $((o+5))" "fuzz_frontends.py 60 $((o+6))"; do
  echo "== $f"; timeout 1500 python scripts/debug/$f < /dev/null 2>&1 | grep -v ": ok$" | tail -8 | cut -c1-260
done
