# A/B of library builds under transformertts_amd/build/ab (development aid): default build first, then each variant
set -e
unset TTTS_LIB
python tools/gemm_ablate.py
for f in transformertts_amd/build/ab/lib_*.so; do
  TTTS_LIB=$PWD/$f python tools/gemm_ablate.py
done
