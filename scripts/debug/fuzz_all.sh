for f in "fuzz_generic.py 60 21" "fuzz_pipeline.py 12 22" "fuzz_stream.py 25 23" "fuzz_train.py 20 24" "fuzz_mfcc.py 20 25" "fuzz_frontends.py 10 26"; do
  echo "== $f"; timeout 250 python scripts/debug/$f < /dev/null 2>&1 | tail -3 | cut -c1-220
done
