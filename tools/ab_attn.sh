# A/B of library builds under transformertts_amd/build/ab on the attention shapes (development aid)
set -e
unset TTTS_LIB
echo default; python tools/attn_bench.py
for f in transformertts_amd/build/ab/lib_*.so; do
  echo $f; TTTS_LIB=$PWD/$f python tools/attn_bench.py
done
